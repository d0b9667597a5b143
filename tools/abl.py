"""Ablation timing helper (dev only): python tools/abl.py <lib.so> [k] [B] -> forward us for each precision."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kgdet_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != '-':
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from kgdet_amd import dcn
k = int(sys.argv[2]) if len(sys.argv) > 2 else 7
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device('cuda:0')
torch.manual_seed(0)
C, H, W = 256, 25, 42
x = torch.randn(B, C, H, W, device=dev)
off = torch.randn(B, 2 * k * k, H, W, device=dev) * 2
w = torch.randn(C, C, k, k, device=dev) * 0.01
shape = dcn._shape(x, w, (1, 1), (k // 2, k // 2), (1, 1), 1, 1)
packed = dcn.pack_weight(w, shape)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
res = {}
for prec in ('split', 'bf16'):
    with dcn.forward_precision(prec):
        for _ in range(3):
            dcn._forward(x, off, None, w, None, shape, packed=packed)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(50):
            dcn._forward(x, off, None, w, None, shape, packed=packed)
        e1.record()
        torch.cuda.synchronize()
    res[prec] = round(e0.elapsed_time(e1) / 50 * 1e3, 1)
print(os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else 'default', k, B, res, flush=True)
