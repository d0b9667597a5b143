"""Are two eager runs of N training steps from the same seed bit-identical?   python tools/determinism_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import build_detector, configs, synthetic
from kgdet_amd.dist import DistOptimizerHook
from kgdet_amd.runner import batch_processor
torch.backends.cudnn.benchmark = False
batch = synthetic.make_batch(2, 'cuda', seed=0)
def run(n):
    cfg = configs.kgdet_r50_fpn()
    torch.manual_seed(0)
    model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-5)
    hook = DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2))
    for _ in range(n):
        out = batch_processor(model, batch); hook.step(model, opt, out['loss'])
    torch.cuda.synchronize()
    return [p.detach().clone() for p in model.parameters()], {n_: p for n_, p in model.named_parameters()}
a, names = run(int(os.environ.get('N', '6')))
b, _ = run(int(os.environ.get('N', '6')))
bad = [(n, float((x - y).abs().max())) for (n, _), x, y in zip(names.items(), a, b) if not torch.equal(x, y)]
print('tensors differing between two identical eager runs: %d of %d' % (len(bad), len(a)))
for n, d in bad[:12]: print('  %-60s %.3e' % (n, d))
