"""grouped grad_weight timing: python tools/bench_gw.py [lib.so]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kgdet_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != '-': _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from kgdet_amd import dcn
dev = torch.device('cuda:0'); torch.manual_seed(0)
B, C, H, W = 2, 256, 25, 42
ks = (3, 5, 7)
xs = [torch.randn(B, C, H, W, device=dev) for _ in range(2)]
offs = [torch.randn(B, 2 * k * k, H, W, device=dev) * 2 for k in ks]
ws = [torch.randn(C, C, k, k, device=dev) * 0.01 for _ in xs for k in ks]
go = [torch.randn(B, 3 * C, H, W, device=dev) for _ in xs]
shapes = []
for i in range(2):
    for j, k in enumerate(ks):
        s = dcn._shape(xs[i], ws[i * 3 + j], (1, 1), (k // 2, k // 2), (1, 1), 1, 1)
        s.out_channel_offset, s.out_channels_total = j * C, 3 * C
        shapes.append(s)
args = ([xs[j // 3] for j in range(6)], [offs[j % 3] for j in range(6)], [go[j // 3] for j in range(6)], ws, shapes)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3): dcn.grad_weights_grouped(*args)
torch.cuda.synchronize(); e0.record()
for _ in range(20): dcn.grad_weights_grouped(*args)
e1.record(); torch.cuda.synchronize()
print(os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else '-', 'grouped grad_weight %.1f us' % (e0.elapsed_time(e1) / 20 * 1e3), flush=True)
