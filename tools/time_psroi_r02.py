"""A/B: the round-2 thread-per-bin / global-atomic PS-RoI kernels (built from git history into
tools/experiments/libpsroi_r02_atomic.so, not part of the product) on the shape of tools/time_psroi.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from kgdet_amd import _lib
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'experiments', 'libpsroi_r02_atomic.so'))
rng = np.random.default_rng(7)
B, C, H, W, P, R = 2, 256, 50, 84, 7, 512
data = torch.from_numpy(rng.normal(size=(B, C, H, W)).astype(np.float32)).cuda()
x1 = rng.uniform(-30, 1250, R); y1 = rng.uniform(-30, 720, R)
rois = torch.from_numpy(np.stack([rng.integers(0, B, R), x1, y1, x1 + rng.uniform(8, 600, R), y1 + rng.uniform(8, 500, R)],
                                 1).astype(np.float32)).cuda()
off = torch.from_numpy((rng.normal(size=(R, 2, P, P)) * 0.5).astype(np.float32)).cuda()
go = torch.randn(R, C, P, P, device='cuda')
s = _lib.PsroiShape()
s.B, s.C, s.H, s.W, s.R, s.out_dim, s.group_size, s.pooled_size, s.part_size, s.sample_per_part = B, C, H, W, R, C, 1, P, P, S
s.no_trans, s.num_classes, s.spatial_scale, s.trans_std = 0, 1, 1 / 16., 0.1
out, cnt = torch.empty_like(go), torch.empty_like(go)
gd, gt = torch.zeros_like(data), torch.zeros_like(off)
st = _lib.current_stream()


def fwd():
    assert L.kgdet_deform_psroi_forward(ctypes.byref(s), _lib.ptr(data), _lib.ptr(rois), _lib.ptr(off), _lib.ptr(out), _lib.ptr(cnt), st) == 0


def bwd():
    gd.zero_(); gt.zero_()
    assert L.kgdet_deform_psroi_backward(ctypes.byref(s), _lib.ptr(go), _lib.ptr(cnt), _lib.ptr(data), _lib.ptr(rois),
                                         _lib.ptr(off), _lib.ptr(gd), _lib.ptr(gt), st) == 0


for name, fn in (('forward', fwd), ('backward (incl. zero fill)', bwd)):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print('r02 atomic kernels S=%d %s: %.1f us' % (S, name, e0.elapsed_time(e1) / 20 * 1e3))
a = gd.clone(); bwd(); torch.cuda.synchronize()
print('bitwise repeatable backward:', bool(torch.equal(a, gd)))
