"""phase breakdown of conv3x3_patch from the trace build (make -C kgdet_amd/csrc VARIANT=ctrace EXTRA=-DKGDET_CONV_TRACE):
KGDET_CONV3X3_PATCH=1 KGDET_LIB=kgdet_amd/libkgdet_hip_ctrace.so python tools/conv_trace.py [B C O H W]  (the trace lives in the 8-wave kernel)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from kgdet_amd import conv1x1 as c1, _lib
B, C, O, H, W = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (2, 128, 128, 100, 168)
x = torch.randn(B, C, H, W, device='cuda'); w = torch.randn(O, C, 3, 3, device='cuda') * 0.05
img = c1._pack(w, False)
for _ in range(5): c1._apply(img, x, O, 9)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (1024 * 8))()
assert _lib.lib().kgdet_debug_read_conv_trace(buf) == 0
t = np.array(buf[:], dtype=np.float64).reshape(1024, 8)
t = t[t[:, 7] > 0]
names = ['barrier', 'frag reads', 'mfma', 'commit', 'lds drain', 'prologue', 'total', 'stages']
print('%d workgroups traced (wave 0, lane 0); s_memtime ticks' % len(t))
for c, n in enumerate(names):
    print('  %-10s mean %9.1f  min %9.1f  max %9.1f   per stage %7.1f' % (n, t[:, c].mean(), t[:, c].min(), t[:, c].max(), (t[:, c] / t[:, 7]).mean()))
# placement: workgroups per CU (HW_ID: cu_id bits 8-11, sh_id 12, se_id 13-15; XCC_ID bits 0-3), start / end time
raw = np.array(buf[:], dtype=np.uint64).reshape(1024, 8)
raw = raw[raw[:, 7] > 0]
hw = raw[:, 4]
xcc = (hw >> np.uint64(32)) & np.uint64(15); cu = (hw >> np.uint64(8)) & np.uint64(15); se = (hw >> np.uint64(13)) & np.uint64(7); sh = (hw >> np.uint64(12)) & np.uint64(1)
key = xcc * np.uint64(1000) + se * np.uint64(100) + sh * np.uint64(50) + cu
u, cnt = np.unique(key, return_counts=True)
print('distinct CUs used: %d; workgroups per CU histogram:' % len(u), dict(zip(*np.unique(cnt, return_counts=True))))
start = raw[:, 5].astype(np.float64); start -= start.min()
print('start times (cycles): min 0 median %.0f max %.0f;  duration median %.0f max %.0f' % (np.median(start), start.max(), np.median(raw[:, 6]), raw[:, 6].max()))
late = start > 5000
print('workgroups starting > 5000 cycles after the first: %d; their mean duration %.0f vs others %.0f' % (late.sum(), raw[late, 6].mean() if late.any() else 0, raw[~late, 6].mean()))
