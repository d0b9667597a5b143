"""forward + backward time of one DeformConv call on a large map: python tools/time_bwd_large.py [H W]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import dcn
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100, 168)
torch.manual_seed(0)
x = torch.randn(2, 256, H, W, device='cuda', requires_grad=True)
off = (torch.randn(2, 18, H, W, device='cuda') * 2).requires_grad_()
w = (torch.randn(256, 256, 3, 3, device='cuda') * 0.05).requires_grad_()
g = torch.randn(2, 256, H, W, device='cuda')
for mode in ('split', 'exact'):
    with dcn.arithmetic(mode):
        for _ in range(3):
            out = dcn.deform_conv(x, off, w, 1, 1, 1)
            out.backward(g)
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        tf = tb = 0.0
        for _ in range(10):
            e0.record(); out = dcn.deform_conv(x, off, w, 1, 1, 1); e1.record(); out.backward(g); e2.record()
            torch.cuda.synchronize()
            tf += e0.elapsed_time(e1); tb += e1.elapsed_time(e2)
        print('%dx%d %s: forward %.3f ms  backward %.3f ms' % (H, W, mode, tf / 10, tb / 10), flush=True)
