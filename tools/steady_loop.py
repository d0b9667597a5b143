"""`steps` training steps of the bench workload, nothing else (for tracing): python tools/steady_loop.py [steps=600]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('MIOPEN_USER_DB_PATH', os.path.join(ROOT, 'kgdet_amd', 'miopen_db', 'train_fp32_b2'))
import time  # noqa: E402

import torch  # noqa: E402

from kgdet_amd import configs, synthetic  # noqa: E402
from kgdet_amd.dist import DistOptimizerHook  # noqa: E402
from kgdet_amd.registry import build_detector  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
dev = torch.device('cuda:0')
cfg = configs.kgdet_r50_fpn()
torch.manual_seed(0)
torch.backends.cudnn.benchmark = True
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).to(dev)
batch = synthetic.make_batch(2, dev, seed=0)
model.train()
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.Adam(params, lr=cfg.optimizer.lr, fused=True)
hook = DistOptimizerHook(grad_clip=dict(cfg.optimizer_config.grad_clip), overlap=True, bucket_size_mb=32)
t0 = time.time()
for i in range(steps):
    losses = model(batch['img'], batch['img_meta'], return_loss=True, gt_bboxes=batch['gt_bboxes'],
                   gt_labels=batch['gt_labels'], gt_keypoints=batch['gt_keypoints'])
    loss = sum(v.float() if torch.is_tensor(v) else sum(x.float() for x in v) for k, v in losses.items() if 'loss' in k)
    hook.step(model, opt, loss)
    if i % 100 == 99:
        torch.cuda.synchronize()
        print('step %d  %.2f ms/step so far' % (i + 1, (time.time() - t0) / (i + 1) * 1e3), flush=True)
torch.cuda.synchronize()
