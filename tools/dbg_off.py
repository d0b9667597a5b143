import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import dcn
torch.manual_seed(0)
dev = 'cuda:0'
B, C, H, W, k = 2, 256, 25, 42, 3
x = torch.randn(B, C, H, W, device=dev)
off = torch.randn(B, 2 * k * k, H, W, device=dev) * 2
w = torch.randn(C, C, k, k, device=dev) * 0.05
go = torch.randn(B, C, H, W, device=dev)
shape = dcn._shape(x, w, (1, 1), (k // 2, k // 2), (1, 1), 1, 1)
packed = dcn.pack_weight(w, shape)
needs = dict(input=True, offset=True, mask=False, weight=False, bias=False)
gi, ref, _, _, _ = dcn._backward(x, off, None, w, None, go, shape, packed, needs)
a = dcn.grad_offset_plane(x, off, w, go, shape, packed)
b = dcn.grad_offset_plane(x, off, w, go, shape, packed)
scale = ref.abs().max().item()
print('equal runs:', torch.equal(a, b), 'max diff runs', (a - b).abs().max().item() / scale)
d = (a - ref).abs() / scale
print('max err', d.max().item(), 'mean', d.mean().item())
bad = d > 1e-4
print('bad frac', bad.float().mean().item())
idx = bad.nonzero()
if idx.numel():
    print('bad b', idx[:, 0].unique().tolist(), 'bad ch', idx[:, 1].unique().tolist()[:20])
    hw = idx[:, 2] * W + idx[:, 3]
    u = hw.unique()
    print('bad hw count', u.numel(), u.tolist()[:64])
    print('hw mod 16 hist', torch.bincount(u % 16, minlength=16).tolist())
    print('hw // 128 hist', torch.bincount(u // 128, minlength=9).tolist())
