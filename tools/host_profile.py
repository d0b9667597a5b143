"""cProfile of the host side of training steps (where does the Python time go?)."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('MIOPEN_USER_DB_PATH', os.path.join(ROOT, 'kgdet_amd', 'miopen_db', 'train_fp32_b2'))
import torch
torch.backends.cudnn.benchmark = True
from kgdet_amd import build_detector, configs, synthetic
from kgdet_amd.dist import DistOptimizerHook
dev = torch.device('cuda:0')
cfg = configs.kgdet_r50_fpn()
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).to(dev).train()
batch = synthetic.make_batch(2, dev, seed=0)
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4, fused=True)
hook = DistOptimizerHook(grad_clip=dict(cfg.optimizer_config.grad_clip))
def step():
    losses = model(batch['img'], batch['img_meta'], return_loss=True, gt_bboxes=batch['gt_bboxes'],
                   gt_labels=batch['gt_labels'], gt_keypoints=batch['gt_keypoints'])
    loss = sum(v.float() if torch.is_tensor(v) else sum(x.float() for x in v) for k, v in losses.items() if 'loss' in k)
    hook.step(model, opt, loss)
for _ in range(6): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(28)
