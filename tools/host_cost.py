"""Host-side cost per call (enqueue only) of the conv wrappers vs F.conv2d, and of bn_act vs the module ops."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from kgdet_amd import conv1x1 as c1, backbone as bb
x = torch.randn(2, 256, 16, 32, device='cuda', requires_grad=True); w = torch.randn(64, 256, 1, 1, device='cuda', requires_grad=True)
gy = torch.randn(2, 64, 16, 32, device='cuda')
bn = torch.nn.BatchNorm2d(64).cuda().eval()
def host(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    t = (time.time() - t0) / n * 1e6; torch.cuda.synchronize(); return t
def fb(conv):
    def f():
        y = conv(x, w); y.backward(gy); x.grad = None; w.grad = None
    return f
print('fwd      F.conv2d %.1f us   conv_split %.1f us' % (host(lambda: F.conv2d(x, w)), host(lambda: c1.conv_split(x, w))))
print('fwd+bwd  F.conv2d %.1f us   conv_split %.1f us' % (host(fb(F.conv2d)), host(fb(c1.conv_split))))
y0 = torch.randn(2, 64, 16, 32, device='cuda', requires_grad=True)
def bnf(fused):
    def f():
        z = bb.frozen_bn_act(y0, bn, None, True) if fused else F.relu(bn(y0))
        z.backward(gy); y0.grad = None; bn.zero_grad()
    return f
print('bn+relu fwd+bwd  torch %.1f us   fused %.1f us' % (host(bnf(False)), host(bnf(True))))
