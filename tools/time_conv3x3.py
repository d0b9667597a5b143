"""Kernel time of the 3x3 (or 1x1) forward convolution alone (packed image ready): HIP events around 200 back-to-back calls.
python tools/time_conv3x3.py [k1]   (KGDET_LIB=... selects an experiment build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import conv1x1 as c1
k = 1 if 'k1' in sys.argv else 3
shapes = [(2, 128, 128, 100, 168)] if "one" in sys.argv else [(2, 256, 256, 25, 42), (2, 512, 512, 25, 42), (2, 256, 256, 50, 84), (2, 128, 128, 100, 168), (2, 64, 64, 200, 336)]
if k == 1:   # the bottleneck 1x1 shapes of a [2, 3, 800, 1344] step (+ variants with fewer pixel tiles)
    shapes = [(2, 256, 64, 200, 336), (2, 64, 256, 200, 336), (2, 256, 128, 200, 336), (2, 512, 128, 100, 168), (2, 512, 128, 96, 168),
              (2, 128, 512, 100, 168), (2, 128, 512, 96, 168), (2, 512, 256, 100, 168), (2, 1024, 256, 50, 84), (2, 256, 1024, 50, 84),
              (2, 1024, 512, 50, 84), (2, 2048, 512, 25, 42), (2, 512, 2048, 25, 42), (2, 1024, 2048, 25, 42)]
for B, C, O, H, W in shapes:
    x = torch.randn(B, C, H, W, device='cuda'); w = torch.randn(O, C, k, k, device='cuda') * 0.05
    img = c1._pack(w, False)
    for _ in range(10): c1._apply(img, x, O, k * k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    e0.record()
    for _ in range(n): c1._apply(img, x, O, k * k)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    gf = 2 * B * C * O * H * W * k * k / 1e9
    print('C=%4d O=%4d %3dx%-3d  %6.1f us  %5.1f TF/s (%.2f of 833)  %5.2f TB/s  tiles %d' % (C, O, H, W, us, gf / us * 1e3, gf / us * 1e3 / 833, 4e-6 * B * H * W * (C + O) / us, B * ((O + 127) // 128) * ((H * W + 127) // 128)))
