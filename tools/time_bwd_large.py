"""forward + backward time of one DeformConv call on a large map: python tools/time_bwd_large.py [H W [hot]]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import dcn
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100, 168)
torch.manual_seed(0)
x = torch.randn(2, 256, H, W, device='cuda', requires_grad=True)
off = torch.randn(2, 18, H, W, device='cuda') * 2
if len(sys.argv) > 3 and sys.argv[3] == 'hot':      # every sample next to one of 17 points per image (offsets trained onto key points)
    ys, xs = torch.meshgrid(torch.arange(H, device='cuda', dtype=torch.float32), torch.arange(W, device='cuda', dtype=torch.float32), indexing='ij')
    pts = torch.rand(2, 17, 2, device='cuda') * torch.tensor([H - 1.0, W - 1.0], device='cuda')
    for t in range(9):
        which = torch.randint(0, 17, (2, H, W), device='cuda')
        ty = torch.gather(pts[:, :, 0], 1, which.view(2, -1)).view(2, H, W) + 0.3 * torch.randn(2, H, W, device='cuda')
        tx = torch.gather(pts[:, :, 1], 1, which.view(2, -1)).view(2, H, W) + 0.3 * torch.randn(2, H, W, device='cuda')
        off[:, 2 * t] = ty - (ys - 1 + t // 3)
        off[:, 2 * t + 1] = tx - (xs - 1 + t % 3)
off = off.requires_grad_()
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
