"""Host-side enqueue cost of a KGDet training step: the same step on a 256 x 320 image (GPU work ~1/15) -- when the step
time there is close to the full-size step, the full-size step is launch-bound in places.   python tools/host_step_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import build_detector, configs, synthetic
from kgdet_amd.dist import DistOptimizerHook
SERIAL = os.environ.get('CONFIG', 'kgdet') == 'serial'      # CONFIG=serial: BASELINE config 5 (five-level serial head, SGD)
cfg = configs.reppoints_kp_r50_fpn() if SERIAL else configs.kgdet_r50_fpn()
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
opt = (torch.optim.SGD(model.parameters(), lr=1e-5, momentum=0.9, fused=True) if SERIAL
       else torch.optim.Adam(model.parameters(), lr=1e-5, fused=True))
hook = DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2), overlap=True, bucket_size_mb=32)
for shape in ((256, 320), (800, 1344)):
    batch = synthetic.make_batch(2, 'cuda', seed=0, img_shape=(shape[0], shape[1], 3), pad_shape=(shape[0], shape[1], 3))
    for b in batch['gt_bboxes']:
        b.mul_(min(shape) / 1344.0)
    for k in batch['gt_keypoints']:
        k[:, :, :2].mul_(min(shape) / 1344.0)
    def step():
        losses = model(batch['img'], batch['img_meta'], return_loss=True, gt_bboxes=batch['gt_bboxes'],
                       gt_labels=batch['gt_labels'], gt_keypoints=batch['gt_keypoints'])
        loss = sum(v if torch.is_tensor(v) else sum(v) for k, v in losses.items() if 'loss' in k)
        hook.step(model, opt, loss)
    for _ in range(8): step()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(20): step()
        t1 = time.time()            # host done enqueueing
        torch.cuda.synchronize(); t2 = time.time()
        ts.append(((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
    print(shape, 'host enqueue %.2f ms/step, step %.2f ms' % sorted(ts)[1], flush=True)
