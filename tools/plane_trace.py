"""phase breakdown of the plane forward kernel from the trace build (make VARIANT=trace EXTRA=-DKGDET_PLANE_TRACE):
KGDET_LIB=kgdet_amd/libkgdet_hip_trace.so python tools/plane_trace.py [B] [prec]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from kgdet_amd import dcn, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
prec = sys.argv[2] if len(sys.argv) > 2 else 'split'
dev = torch.device('cuda:0')
torch.manual_seed(0)
C, H, W = 256, 25, 42
xs = [torch.randn(B, C, H, W, device=dev) for _ in range(2)]
ks = (3, 5, 7)
offs = [torch.randn(B, 2 * k * k, H, W, device=dev) * 2 for k in ks]
ws = [[torch.randn(C, C, k, k, device=dev) * 0.01 for k in ks] for _ in xs]
with torch.no_grad(), dcn.forward_precision(prec):
    for _ in range(3):
        dcn.deform_conv_cat_multi(xs, offs, ws, [k // 2 for k in ks])
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (256 * 2 * 10))()
assert _lib.lib().kgdet_debug_read_plane_trace(buf) == 0
t = np.array(buf[:], dtype=np.float64).reshape(256, 2, 10) / 100.0     # s_memtime ticks at 100 MHz -> us
names = ['pro_wait0', 'pro_load', 'pro_wait1', 'pro_sample', 'stage_work', 'stage_wait', 'epilogue', 'total', 'stages', 'segments']
for role, rn in ((0, 'consumer wave 0'), (1, 'producer wave 8')):
    print(rn)
    for c, nme in enumerate(names):
        v = t[:, role, c] * (100.0 if c >= 8 else 1.0)
        print('  %-11s mean %8.1f  min %8.1f  max %8.1f' % (nme, v.mean(), v.min(), v.max()))
seg = t[:, 0, 9] * 100
for lo, hi, tag in ((0, 6, '<=5 segments (7x7)'), (6, 12, '6-11 (5x5)'), (12, 99, '>=12 (3x3)')):
    m = (seg >= lo) & (seg < hi)
    if m.any():
        print(tag, 'workgroups', int(m.sum()), 'total us mean %.1f' % t[m, 0, 7].mean(),
              ' '.join('%s %.1f' % (names[c], t[m, 0, c].mean()) for c in range(7)))
