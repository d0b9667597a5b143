import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from kgdet_amd import conv1x1
torch.manual_seed(0)
for (B, C, O, H, W, k) in [(2, 256, 256, 32, 40, 3), (2, 256, 256, 16, 20, 3), (2, 256, 256, 8, 10, 3), (2, 256, 256, 4, 5, 3),
                        (2, 256, 256, 2, 3, 3), (2, 256, 256, 25, 42, 3), (2, 256, 256, 100, 168, 3), (2, 256, 256, 4, 5, 1),
                        (2, 256, 256, 6, 6, 3), (2, 256, 256, 3, 4, 3), (1, 256, 256, 4, 5, 3), (2, 64, 64, 4, 5, 3)]:
    x = torch.randn(B, C, H, W, device='cuda', dtype=torch.float64)
    w = torch.randn(O, C, k, k, device='cuda', dtype=torch.float64) * 0.05
    gy = torch.randn(B, O, H, W, device='cuda', dtype=torch.float64)
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    yr = F.conv2d(xr, wr, padding=k // 2)
    yr.backward(gy)
    xs, ws = x.float().requires_grad_(), w.float().requires_grad_()
    assert conv1x1.applicable(xs, ws, (1, 1), (k // 2, k // 2))
    ys = conv1x1.conv_split(xs, ws)
    ys.backward(gy.float())
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
    print((B, C, O, H, W, k), 'fwd %.2e gx %.2e gw %.2e' % (rel(ys, yr), rel(xs.grad, xr.grad), rel(ws.grad, wr.grad)), flush=True)
