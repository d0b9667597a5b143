"""Can the KGDet training step be captured as one HIP graph?  Eager vs replayed step time.   python tools/graph_train_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import build_detector, configs, synthetic
from kgdet_amd.dist import DistOptimizerHook
SERIAL = os.environ.get('CONFIG', 'kgdet') == 'serial'      # CONFIG=serial: BASELINE config 5
cfg = configs.reppoints_kp_r50_fpn() if SERIAL else configs.kgdet_r50_fpn()
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
capturable = os.environ.get('KGDET_FUSED_CLIP_ADAM', '1') == '0'
opt = (torch.optim.SGD(model.parameters(), lr=1e-5, momentum=0.9, fused=True) if SERIAL
       else torch.optim.Adam(model.parameters(), lr=1e-5, fused=True, capturable=capturable))
hook = DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2), overlap=True, bucket_size_mb=32)
batch = synthetic.make_batch(2, 'cuda', seed=0)
STAGE = os.environ.get('STAGE', 'full')
def step():
    if STAGE != 'full':
        losses = model(batch['img'], batch['img_meta'], return_loss=True, gt_bboxes=batch['gt_bboxes'],
                       gt_labels=batch['gt_labels'], gt_keypoints=batch['gt_keypoints'])
        loss = sum(v if torch.is_tensor(v) else sum(v) for k, v in losses.items() if 'loss' in k)
        if STAGE == 'bwd':
            opt.zero_grad()
            loss.backward()
        return loss
    losses = model(batch['img'], batch['img_meta'], return_loss=True, gt_bboxes=batch['gt_bboxes'],
                   gt_labels=batch['gt_labels'], gt_keypoints=batch['gt_keypoints'])
    loss = sum(v if torch.is_tensor(v) else sum(v) for k, v in losses.items() if 'loss' in k)
    hook.step(model, opt, loss)
    return loss
def timed(fn, n=40):
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(8): step()
torch.cuda.current_stream().wait_stream(side)
print('eager %.2f ms/step' % timed(step), flush=True)
if hasattr(hook._fused, 'enable_device_schedule'):
    hook._fused.enable_device_schedule(opt)
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    static_loss = step()
print('captured', flush=True)
for _ in range(5): g.replay()
torch.cuda.synchronize()
print('loss after replays', float(static_loss), flush=True)
print('graph %.2f ms/step' % timed(g.replay), flush=True)
print('graph %.2f ms/step' % timed(g.replay), flush=True)
