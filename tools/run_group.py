"""Run the grouped head forward (2 feature maps x 3x3/5x5/7x7) a few times (profiling target): python tools/run_group.py [B] [iters] [prec]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import dcn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
prec = sys.argv[3] if len(sys.argv) > 3 else 'split'
dev = torch.device('cuda:0')
torch.manual_seed(0)
C, H, W = 256, 25, 42
xs = [torch.randn(B, C, H, W, device=dev) for _ in range(2)]
ks = (3, 5, 7)
offs = [torch.randn(B, 2 * k * k, H, W, device=dev) * 2 for k in ks]
ws = [[torch.randn(C, C, k, k, device=dev) * 0.01 for k in ks] for _ in xs]
with torch.no_grad(), dcn.forward_precision(prec):
    for _ in range(iters):
        dcn.deform_conv_cat_multi(xs, offs, ws, [k // 2 for k in ks])
torch.cuda.synchronize()
