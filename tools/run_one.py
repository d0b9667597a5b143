"""Run one deformable-conv pass repeatedly (profiling target): python tools/run_one.py {fwd|bwd_in|bwd_w} k [B] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kgdet_amd import dcn

which, k = sys.argv[1], int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device('cuda:0')
torch.manual_seed(0)
C, H, W, K = 256, 25, 42, k * k
x = torch.randn(B, C, H, W, device=dev)
off = torch.randn(B, 2 * K, H, W, device=dev) * 2
w = torch.randn(C, C, k, k, device=dev) * 0.01
go = torch.randn(B, C, H, W, device=dev)
shape = dcn._shape(x, w, (1, 1), (k // 2, k // 2), (1, 1), 1, 1)
packed = dcn.pack_weight(w, shape)
needs = dict(input=which == 'bwd_in', offset=which == 'bwd_in', mask=False, weight=which == 'bwd_w', bias=False)
for _ in range(iters):
    if which == 'fwd':
        dcn._forward(x, off, None, w, None, shape, packed=packed)
    else:
        dcn._backward(x, off, None, w, None, go, shape, packed, needs)
torch.cuda.synchronize()
