"""Forward timing helper (dev only): python tools/abl.py <lib.so|-> [k|group] [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kgdet_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != '-':
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from kgdet_amd import dcn
mode = sys.argv[2] if len(sys.argv) > 2 else '7'
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device('cuda:0')
torch.manual_seed(0)
C, H, W = 256, 25, 42
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
res = {}
if mode == 'group':
    xs = [torch.randn(B, C, H, W, device=dev) for _ in range(2)]
    ks = (3, 5, 7)
    offs = [torch.randn(B, 2 * k * k, H, W, device=dev) * 2 for k in ks]
    ws = [[torch.randn(C, C, k, k, device=dev) * 0.01 for k in ks] for _ in xs]
    pads = [k // 2 for k in ks]
    dcn._pack_cache_enabled = True if hasattr(dcn, '_pack_cache_enabled') else None
    def run():
        with torch.no_grad():
            dcn.deform_conv_cat_multi(xs, offs, ws, pads)
else:
    k = int(mode)
    x = torch.randn(B, C, H, W, device=dev)
    off = torch.randn(B, 2 * k * k, H, W, device=dev) * 2
    w = torch.randn(C, C, k, k, device=dev) * 0.01
    shape = dcn._shape(x, w, (1, 1), (k // 2, k // 2), (1, 1), 1, 1)
    packed = dcn.pack_weight(w, shape)
    def run():
        dcn._forward(x, off, None, w, None, shape, packed=packed)
for prec in ('split', 'bf16'):
    with dcn.forward_precision(prec):
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(30):
            run()
        e1.record()
        torch.cuda.synchronize()
    res[prec] = round(e0.elapsed_time(e1) / 30 * 1e3, 1)
print(os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else 'default', mode, B, res, flush=True)
