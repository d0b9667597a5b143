"""phase breakdown of the column-wave forward kernel from the trace build (make VARIANT=cwtrace EXTRA=-DKGDET_CW_TRACE):
KGDET_DCN_CW=1 KGDET_LIB=kgdet_amd/libkgdet_hip_cwtrace.so python tools/cw_trace.py [B]
per problem class and wave: shader cycles per STAGE spent at the top of an iteration, in the main block, at the barrier"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from kgdet_amd import dcn, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device('cuda:0')
torch.manual_seed(0)
C, H, W = 256, 25, 42
xs = [torch.randn(B, C, H, W, device=dev) for _ in range(2)]
ks = (3, 5, 7)
offs = [torch.randn(B, 2 * k * k, H, W, device=dev) * 2 for k in ks]
ws = [[torch.randn(C, C, k, k, device=dev) * 0.01 for k in ks] for _ in xs]
with torch.no_grad(), dcn.forward_precision('split'):
    for _ in range(3):
        dcn.deform_conv_cat_multi(xs, offs, ws, [k // 2 for k in ks])
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (256 * 4 * 12))()
assert _lib.lib().kgdet_debug_read_cw_trace(buf) == 0
t = np.array(buf[:], dtype=np.float64).reshape(256, 4, 12)
its = t[:, 0, 5]
for lo, hi, tag in ((1, 150, '3x3 (144 stages)'), (190, 198, '7x7 (196)'), (198, 260, '5x5 (200)')):
    m = (its >= lo) & (its < hi)
    if not m.any():
        continue
    print('%s: %d workgroups, total %.0f cycles' % (tag, int(m.sum()), t[m, 0, 6].mean()))
    for w in range(4):
        n = t[m, w, 5].mean()
        print('  wave %d per stage: top %.0f  main %.0f  barrier %.0f | prologue %.0f epilogue %.0f' % (
            w, t[m, w, 7].mean() / n, t[m, w, 1].mean() / n, t[m, w, 2].mean() / n, t[m, w, 3].mean(), t[m, w, 4].mean()))
