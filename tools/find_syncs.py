"""List host<->device synchronisation points of a training step (torch sync debug mode = warn)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import build_detector, configs, synthetic
from kgdet_amd.dist import DistOptimizerHook
dev = torch.device('cuda:0')
cfg = configs.reppoints_kp_r50_fpn(soft_nms=True) if os.environ.get('CONFIG') == 'serial' else configs.kgdet_r50_fpn()
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).to(dev)
batch = synthetic.make_batch(2, dev, seed=0)
model.train()
opt = (torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=5e-3, momentum=0.9, weight_decay=1e-4, fused=True)
       if os.environ.get('CONFIG') == 'serial' else torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4, fused=True))
hook = DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2))
def step():
    losses = model(batch['img'], batch['img_meta'], return_loss=True, gt_bboxes=batch['gt_bboxes'],
                   gt_labels=batch['gt_labels'], gt_keypoints=batch['gt_keypoints'])
    loss = sum(sum(v) if isinstance(v, (list, tuple)) else v for v in losses.values())
    hook.step(model, opt, loss)
for _ in range(3): step()
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode('warn')
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    step()
torch.cuda.set_sync_debug_mode('default')
import traceback
print('sync warnings:', len(w))
seen = {}
for x in w:
    key = (x.filename, x.lineno)
    seen[key] = seen.get(key, 0) + 1
for k, v in seen.items(): print(v, k)
