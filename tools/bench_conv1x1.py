"""1x1 convolution micro-benchmark: split-bf16 GEMM kernels vs MIOpen fp32, forward / grad_input / grad_weight."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from kgdet_amd import conv1x1 as c1
torch.backends.cudnn.benchmark = 'nofind' not in sys.argv
k = 3 if 'k3' in sys.argv else 1
shapes = [(2, 64, 64, 200, 336), (2, 128, 128, 100, 168), (2, 256, 256, 50, 84), (2, 512, 512, 25, 42)] if k == 3 else [(2, 256, 64, 200, 336), (2, 64, 256, 200, 336), (2, 512, 128, 100, 168), (2, 128, 512, 100, 168),
          (2, 1024, 256, 50, 84), (2, 256, 1024, 50, 84), (2, 2048, 512, 25, 42), (2, 512, 2048, 25, 42)]
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e6
for B, C, O, H, W in shapes:
    x = torch.randn(B, C, H, W, device='cuda'); w = torch.randn(O, C, k, k, device='cuda') * 0.05
    gy = torch.randn(B, O, H, W, device='cuda')
    ref = F.conv2d(x.double(), w.double(), padding=k // 2)
    got = c1.conv1x1(x, w)
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    err32 = ((F.conv2d(x, w, padding=k // 2) - ref).abs().max() / ref.abs().max()).item()
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    def mi():
        y = F.conv2d(xr, wr, padding=k // 2); y.backward(gy); xr.grad = None; wr.grad = None
    def mine():
        y = c1.conv1x1(xr, wr); y.backward(gy); xr.grad = None; wr.grad = None
    gf = 2 * B * C * O * H * W * k * k / 1e9
    tf_mi, tf_me = t(lambda: F.conv2d(x, w, padding=k // 2)), t(lambda: c1.conv1x1(x, w))
    print('C=%4d O=%4d %3dx%-3d  fwd miopen %6.1f us (%5.1f TF)  split %6.1f us (%5.1f TF)   fwd+bwd miopen %6.1f  split %6.1f   err %.1e (fp32 conv %.1e)'
          % (C, O, H, W, tf_mi, gf / tf_mi * 1e3, tf_me, gf / tf_me * 1e3, t(mi), t(mine), err, err32))
