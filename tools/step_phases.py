"""Where does a training step's wall time go?  python tools/step_phases.py  (GPU)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import build_detector, configs, synthetic
from kgdet_amd.dist import DistOptimizerHook
dev = torch.device('cuda:0')
cfg = configs.kgdet_r50_fpn()
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).to(dev)
batch = synthetic.make_batch(2, dev, seed=0)
model.train()
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4, fused=True)
hook = DistOptimizerHook(grad_clip=dict(cfg.optimizer_config.grad_clip))
def sync(): torch.cuda.synchronize()
acc = {}
def run(timed):
    t = {}
    sync(); t0 = time.time()
    x = model.extract_feat(batch['img'])
    outs = model.bbox_head(x, batch['img_meta'])
    t['fwd_enqueue'] = time.time() - t0
    if timed: sync()
    t['fwd_done'] = time.time() - t0
    loss_inputs = outs + (batch['gt_bboxes'], batch['gt_labels'], batch['gt_keypoints'], batch['img_meta'], model.train_cfg)
    losses = model.bbox_head.loss(*loss_inputs, gt_bboxes_ignore=None)
    t['loss_enqueue'] = time.time() - t0
    if timed: sync()
    t['loss_done'] = time.time() - t0
    loss = sum(sum(v) if isinstance(v, (list, tuple)) else v for v in losses.values())
    opt.zero_grad()
    loss.backward()
    t['bwd_enqueue'] = time.time() - t0
    if timed: sync()
    t['bwd_done'] = time.time() - t0
    hook.clip_grads(model.parameters())
    opt.step()
    t['opt_enqueue'] = time.time() - t0
    sync()
    t['opt_done'] = time.time() - t0
    return t
for _ in range(5): run(False)
for timed in (True, False):
    tot = {}
    for _ in range(10):
        t = run(timed)
        for k, v in t.items(): tot[k] = tot.get(k, 0) + v
    print('phase-synced' if timed else 'free-running', {k: round(v / 10 * 1e3, 2) for k, v in tot.items()})
