"""forward + ONE backward product of a head stage's grouped DeformConv, as bench.py's dcn_backward_products_live runs it (profiling
target: which launches are inside a live figure): python tools/run_bwd_phase.py <grad_weight|grad_input|grad_offset|forward> [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import _lib, dcn
which = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda:0')
g = torch.Generator(device='cpu').manual_seed(0)
B, C, H, W = 2, 256, 25, 42
ks = (3, 5, 7)
xs = [torch.randn(B, C, H, W, generator=g).to(dev) for _ in range(2)]
offs = [(torch.randn(B, 2 * k * k, H, W, generator=g) * 2).to(dev) for k in ks]
ws = [[(torch.randn(C, C, k, k, generator=g) * 0.01).to(dev) for k in ks] for _ in xs]
pads = [k // 2 for k in ks]
for t in xs + offs:
    t.requires_grad_(which in ('grad_input', 'grad_offset'))
for t in [w for wl in ws for w in wl]:
    t.requires_grad_(which in ('grad_weight', 'forward'))
dcn.MEASUREMENT = True
_lib.check(_lib.lib().kgdet_set_option(3, {'grad_weight': 0, 'forward': 0, 'grad_input': 1, 'grad_offset': 2}[which]), 'opt')
gos = None
for _ in range(iters):
    outs = dcn.deform_conv_cat_multi(xs, offs, ws, pads)
    if gos is None:
        gos = [torch.randn_like(o) for o in outs]
    if which != 'forward':
        torch.autograd.backward(outs, gos)
        for t in xs + offs + [w for wl in ws for w in wl]:
            t.grad = None
torch.cuda.synchronize()
_lib.lib().kgdet_set_option(3, 0)
