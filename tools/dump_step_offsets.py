"""What regime are the deformable backward kernels in during the bench's steady training state?  Trains the bench's model (KGDet
R50-FPN, Adam, the synthetic batch) for N steps, takes the offset tensors of the last step's grouped deformable calls and prints,
per call and kernel size, the distribution of contributions per (image, tap, input cell) -- the quantity that decides which path a
cell's sum takes in grad_input (<= 8: inline records; 9 .. 64: a wave per cell; > 64: a column of dcn_hot_gemm).
python tools/dump_step_offsets.py [steps]      (GPU box; saves gpurun_out/step_offsets.npz)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
from kgdet_amd import configs, synthetic, dcn
from kgdet_amd.dist import DistOptimizerHook
from kgdet_amd.registry import build_detector

dev = torch.device('cuda:0')
cfg = configs.kgdet_r50_fpn()
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).to(dev)
batch = synthetic.make_batch(2, dev, seed=0)
model.train()
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.Adam(params, lr=cfg.optimizer.lr, fused=True)
hook = DistOptimizerHook(grad_clip=dict(cfg.optimizer_config.grad_clip), overlap=True, bucket_size_mb=32)
captured = []
orig = dcn.deform_conv_cat_multi


def spy(xs, offsets, weights, paddings, relu=True):
    if spy.on:
        captured.append([o.detach().float().cpu().numpy() for o in offsets])
    return orig(xs, offsets, weights, paddings, relu)


spy.on = False
dcn.deform_conv_cat_multi = spy
import kgdet_amd.heads as heads
heads.dcn.deform_conv_cat_multi = spy
for it in range(steps):
    spy.on = it == steps - 1
    losses = model(batch['img'], batch['img_meta'], return_loss=True, gt_bboxes=batch['gt_bboxes'], gt_labels=batch['gt_labels'],
                   gt_keypoints=batch['gt_keypoints'])
    loss = sum(v.float() if torch.is_tensor(v) else sum(x.float() for x in v) for k, v in losses.items() if 'loss' in k)
    hook.step(model, opt, loss)
torch.cuda.synchronize()
print('steps', steps, 'loss', float(loss), 'grouped calls in the last step', len(captured))
out = {}
for ci, offs in enumerate(captured):
    for o in offs:
        B, K2, H, W = o.shape
        K = K2 // 2
        k = int(round(K ** 0.5))
        out['call%d_k%d' % (ci, k)] = o
        gy, gx = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
        hist = np.zeros(6, dtype=np.int64)     # cells with n in [1, 8], [9, 16], [17, 64], [65, 256], [257, ...]; contributions in cells > 8
        contrib_by = np.zeros(5, dtype=np.int64)
        for b in range(B):
            for t in range(K):
                y = gy - k // 2 + t // k + o[b, 2 * t]
                x = gx - k // 2 + t % k + o[b, 2 * t + 1]
                y0, x0 = np.floor(y).astype(np.int64), np.floor(x).astype(np.int64)
                cnt = np.zeros(H * W, dtype=np.int64)
                for dy in (0, 1):
                    for dx in (0, 1):
                        yy, xx = y0 + dy, x0 + dx
                        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W) & (y > -1) & (y < H) & (x > -1) & (x < W)
                        np.add.at(cnt, (yy * W + xx)[ok], 1)
                for i, (lo, hi) in enumerate(((1, 8), (9, 16), (17, 64), (65, 256), (257, 1 << 30))):
                    m = (cnt >= lo) & (cnt <= hi)
                    hist[i] += m.sum()
                    contrib_by[i] += cnt[m].sum()
        tot = contrib_by.sum()
        print('call %d  %dx%d  |offset| mean %.2f max %.1f   cells by contributions  1-8: %d  9-16: %d  17-64: %d  65-256: %d  >256: %d'
              '   share of contributions: %s' % (ci, k, k, np.abs(o).mean(), np.abs(o).max(), hist[0], hist[1], hist[2], hist[3], hist[4],
                                                 ' '.join('%.3f' % (c / max(tot, 1)) for c in contrib_by)))
os.makedirs('gpurun_out', exist_ok=True)
np.savez_compressed('gpurun_out/step_offsets.npz', **out)
