"""phase breakdown of the plane forward kernel from the trace build (make VARIANT=trace EXTRA=-DKGDET_PLANE_TRACE):
KGDET_LIB=kgdet_amd/libkgdet_hip_trace.so python tools/plane_trace.py [B] [prec]
per wave (0-7 consumers, 8-15 producers) and per problem class: hundreds of shader cycles spent in each phase"""
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
buf = (ctypes.c_ulonglong * (256 * 16 * 10))()
assert _lib.lib().kgdet_debug_read_plane_trace(buf) == 0
t = np.array(buf[:], dtype=np.float64).reshape(256, 16, 10) / 100.0     # s_memtime ticks = shader cycles; / 100
names = ['mid_wait', 'pro_load', 'pro_wait1', 'pro_sample', 'stage_work', 'stage_wait', 'epilogue', 'total', 'stages', 'segments']
seg = t[:, 0, 9] * 100
classes = [(0, 6, '7x7'), (6, 12, '5x5'), (12, 99, '3x3')]
for lo, hi, tag in classes:
    m = (seg >= lo) & (seg < hi)
    if not m.any():
        continue
    print('%s: %d workgroups, %d stages, %d segments' % (tag, int(m.sum()), int(t[m, 0, 8].mean() * 100), int(seg[m].mean())))
    print('  wave ' + ' '.join('%10s' % n for n in names[:8]))
    for w in range(16):
        print('  %4d ' % w + ' '.join('%10.1f' % t[m, w, c].mean() for c in range(8)))
