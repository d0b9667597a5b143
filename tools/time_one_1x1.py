"""forward time of one 1x1 split convolution:  python tools/time_one_1x1.py B C O H W   (KGDET_CONV_KS / KGDET_CONV_NW force the plan)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import conv1x1 as c1
B, C, O, H, W = [int(a) for a in sys.argv[1:6]]
x = torch.randn(B, C, H, W, device='cuda'); w = torch.randn(O, C, 1, 1, device='cuda') * 0.05
img = c1._pack(w, False)
def f(): return c1._apply(img, x, O, 1)
for _ in range(10): f()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(100): f()
torch.cuda.synchronize(); t = (time.time() - t0) / 100 * 1e6
print('KS=%s NW=%s  %.1f us  %.1f TFLOP/s' % (os.environ.get('KGDET_CONV_KS', '-'), os.environ.get('KGDET_CONV_NW', '-'), t, 2.0 * B * C * O * H * W / t / 1e6))
