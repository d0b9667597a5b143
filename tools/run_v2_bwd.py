"""one modulated (v2) DeformConv forward + backward on the KGDet head shape (kernel-trace target): which kernels run"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import dcn
torch.manual_seed(0)
dev = torch.device('cuda:0')
N, C, H, W, k = 2, 256, 25, 42, 3
x = torch.randn(N, C, H, W, device=dev, requires_grad=True)
off = (torch.randn(N, 2 * k * k, H, W, device=dev) * 2).requires_grad_()
m = torch.rand(N, k * k, H, W, device=dev).requires_grad_()
w = (torch.randn(C, C, k, k, device=dev) * 0.01).requires_grad_()
b = torch.zeros(C, device=dev, requires_grad=True)
for _ in range(3):
    out = dcn.modulated_deform_conv(x, off, m, w, b, 1, 1, 1, 1, 1)
    out.backward(torch.randn_like(out))
torch.cuda.synchronize()
