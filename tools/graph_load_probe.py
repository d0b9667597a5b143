"""GraphedTrainStep.load() (eager copies into the graph's input buffers) between back-to-back replays, KGDet config.
   python tools/graph_load_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import build_detector, configs, synthetic
from kgdet_amd.dist import DistOptimizerHook
from kgdet_amd.runner import GraphedTrainStep
cfg = configs.kgdet_r50_fpn()
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
opt = torch.optim.Adam(model.parameters(), lr=1e-5)
hook = DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2))
batch = synthetic.make_batch(2, 'cuda', seed=0)
batch2 = synthetic.make_batch(2, 'cuda', seed=0)
gs = GraphedTrainStep(model, opt, hook, batch, warmup=3)
print('captured', flush=True)
for i in range(100):
    gs.load(batch2)
    out = gs.step()
    if i % 20 == 0:
        torch.cuda.synchronize(); print('replay', i, float(out['loss'].detach()), flush=True)
torch.cuda.synchronize(); print('done', flush=True)
