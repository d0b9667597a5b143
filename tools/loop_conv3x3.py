"""loops one 3x3 convolution for ~12 s (for tools/power_probe.sh): python tools/loop_conv3x3.py [B C O H W]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import conv1x1 as c1
B, C, O, H, W = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (4, 128, 128, 96, 168)
x = torch.randn(B, C, H, W, device='cuda'); w = torch.randn(O, C, 3, 3, device='cuda') * 0.05
img = c1._pack(w, False)
t0 = time.time(); n = 0
while time.time() - t0 < 12:
    for _ in range(500): c1._apply(img, x, O, 9)
    torch.cuda.synchronize(); n += 500
print('%.1f us per call' % ((time.time() - t0) / n * 1e6))
