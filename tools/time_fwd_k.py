"""Forward time per stage of the plane kernel for kernel sizes whose tap count is / is not a multiple of the group size (4):
2 maps x three k x k kernels as one grouped launch.   python tools/time_fwd_k.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import dcn
dev = torch.device('cuda:0')
torch.manual_seed(0)
B, C, H, W = 2, 256, 25, 42
for k in (3, 4, 5, 6, 7, 8):
    p = k // 2
    Ho, Wo = H + 2 * p - k + 1, W + 2 * p - k + 1
    xs = [torch.randn(B, C, H, W, device=dev) for _ in range(2)]
    offs = [torch.randn(B, 2 * k * k, Ho, Wo, device=dev) * 2 for _ in range(3)]
    ws = [[torch.randn(C, C, k, k, device=dev) * 0.01 for _ in range(3)] for _ in xs]
    with torch.no_grad():
        for _ in range(10):
            dcn.deform_conv_cat_multi(xs, offs, ws, [p] * 3)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                dcn.deform_conv_cat_multi(xs, offs, ws, [p] * 3)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 50 * 1e3)
    t = sorted(ts)[2]
    tiles = B * ((Ho * Wo + 127) // 128)
    stages = 2 * 3 * tiles * 16 * k * k
    print('k=%d K=%2d (K %% 4 = %d)  %7.1f us per launch  %6.1f stages per workgroup  %5.2f us per 100 stages' %
          (k, k * k, (k * k) % 4, t, stages / 256.0, t / (stages / 256.0) * 100))
