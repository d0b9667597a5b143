"""event-timed deformable PS-RoI pooling at the DeformRoIPoolingPack shape: 512 RoIs x 256 channels x 7x7 bins on
[2, 256, 50, 84]; prints us per launch and the algorithmic HBM rate (python tools/time_psroi.py [S] [no_trans] [clustered])"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from kgdet_amd.deform_pool import deform_roi_pooling
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
no_trans = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
rng = np.random.default_rng(7)
B, C, H, W, P, R = 2, 256, 50, 84, 7, 512
data = torch.from_numpy(rng.normal(size=(B, C, H, W)).astype(np.float32)).cuda().requires_grad_()
x1 = rng.uniform(-30, 1250, R); y1 = rng.uniform(-30, 720, R)
rois = torch.from_numpy(np.stack([rng.integers(0, B, R), x1, y1, x1 + rng.uniform(8, 600, R), y1 + rng.uniform(8, 500, R)],
                                 1).astype(np.float32)).cuda()
if len(sys.argv) > 3 and sys.argv[3] == 'clustered':     # proposals as an RPN leaves them: jittered copies around 8 objects per image
    objs = np.stack([rng.uniform(100, 1100, 16), rng.uniform(100, 600, 16), rng.uniform(60, 400, 16), rng.uniform(60, 400, 16)], 1)
    which = rng.integers(0, 16, R)
    cx, cy = objs[which, 0] + rng.normal(0, 12, R), objs[which, 1] + rng.normal(0, 12, R)
    w, h = objs[which, 2] * np.exp(rng.normal(0, 0.15, R)), objs[which, 3] * np.exp(rng.normal(0, 0.15, R))
    rois = torch.from_numpy(np.stack([which % 2, cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).astype(np.float32)).cuda()
off = torch.from_numpy((rng.normal(size=(R, 2, P, P)) * 0.5).astype(np.float32)).cuda().requires_grad_()
go = torch.randn(R, C, P, P, device='cuda')


def run(bwd):
    out = deform_roi_pooling(data, rois, off if not no_trans else off.new_empty(0), 1 / 16., P, C, no_trans, 1, P, S, 0.1)
    if bwd:
        out.backward(go)
        data.grad = None
        off.grad = None


res = {}
for bwd in (False, True):
    for _ in range(5):
        run(bwd)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run(bwd)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    res[bwd] = sorted(ts)[2]
fwd_bytes = data.numel() * 4 + 2 * go.numel() * 4            # map read once, out + count written
bwd_bytes = data.numel() * 4 * 2 + 2 * go.numel() * 4      # grad_out + count read, map read (grad_trans) and written
print('S=%d no_trans=%d: forward %.1f us (%.0f GB/s algorithmic), backward %.1f us (%.0f GB/s algorithmic)' % (
    S, no_trans, res[False], fwd_bytes / res[False] / 1e3, res[True] - res[False], bwd_bytes / (res[True] - res[False]) / 1e3))
