"""phase breakdown of the tap-pair grad_offset kernel from the trace build (make VARIANT=pair_trace EXTRA=-DKGDET_PAIR_TRACE):
KGDET_LIB=kgdet_amd/libkgdet_hip_pair_trace.so python tools/pair_trace.py [B]
per wave (0-3 half 0, 4-7 half 1) and per problem class: hundreds of shader cycles spent in each phase"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from kgdet_amd import dcn, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device('cuda:0')
torch.manual_seed(0)
C, H, W = 256, 25, 42
ks = (3, 5, 7)
xs = [torch.randn(B, C, H, W, device=dev, requires_grad=True) for _ in range(2)]
offs = [(torch.randn(B, 2 * k * k, H, W, device=dev) * 2).requires_grad_() for k in ks]
ws = [[(torch.randn(C, C, k, k, device=dev) * 0.01).requires_grad_() for k in ks] for _ in xs]
gos = None
for _ in range(3):
    outs = dcn.deform_conv_cat_multi(xs, offs, ws, [k // 2 for k in ks])
    if gos is None:
        gos = [torch.randn_like(o) for o in outs]
    torch.autograd.backward(outs, gos)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (256 * 8 * 10))()
assert _lib.lib().kgdet_debug_read_pair_trace(buf) == 0
t = np.array(buf[:], dtype=np.float64).reshape(256, 8, 10) / 100.0
names = ['prologue', 'dma_issue', 'matrix', 'tail', 'vm0_wait', 'barrier', 'plane', 'total']
seg = t[:, 0, 9] * 100
for lo, hi, tag in [(0, 6, '7x7'), (6, 12, '5x5'), (12, 99, '3x3')]:
    m = (seg >= lo) & (seg < hi) & (t[:, 0, 7] > 0)
    if not m.any():
        continue
    print('%s: %d workgroups, %d slots, %d segments' % (tag, int(m.sum()), int(t[m, 0, 8].mean() * 100), int(seg[m].mean())))
    print('  wave ' + ' '.join('%10s' % n for n in names))
    for w in range(8):
        print('  %4d ' % w + ' '.join('%10.1f' % t[m, w, c].mean() for c in range(8)))
