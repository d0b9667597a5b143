"""3x3 grad_weight kernel timing for the training shapes (HIP events)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import conv1x1 as c1, _lib
for B, C, O, H, W in [(2, 128, 128, 100, 168), (2, 256, 256, 50, 84), (2, 512, 512, 25, 42), (2, 256, 256, 25, 42)]:
    x = torch.randn(B, C, H, W, device='cuda'); gy = torch.randn(B, O, H, W, device='cuda')
    L = c1._lib_sizes()
    nbytes = L.kgdet_conv3x3_grad_weight_workspace_bytes(ctypes.c_int64(B), ctypes.c_int32(O), ctypes.c_int32(C), ctypes.c_int32(H), ctypes.c_int32(W))
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda'); gw = torch.empty(O, C, 3, 3, device='cuda')
    def k():
        _lib.check(L.kgdet_conv3x3_grad_weight(_lib.ptr(gy), _lib.ptr(x), _lib.ptr(gw), ctypes.c_int64(B), ctypes.c_int32(O), ctypes.c_int32(C),
                                               ctypes.c_int32(H), ctypes.c_int32(W), _lib.ptr(ws), ctypes.c_size_t(nbytes), _lib.current_stream()), 'gw')
    for _ in range(5): k()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): k()
    e1.record(); torch.cuda.synchronize()
    print('C=%4d O=%4d %3dx%-3d  wgrad3x3 %6.1f us  (ws %.1f MB)' % (C, O, H, W, e0.elapsed_time(e1) / 50 * 1e3, nbytes / 1e6))
