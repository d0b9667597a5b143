"""grad_weight kernel timing for the training shapes (HIP events)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import conv1x1 as c1
shapes = [(2, 256, 128, 200, 336), (2, 512, 128, 100, 168), (2, 128, 512, 100, 168), (2, 512, 256, 100, 168),
          (2, 1024, 256, 50, 84), (2, 256, 1024, 50, 84), (2, 1024, 512, 50, 84), (2, 2048, 512, 25, 42), (2, 512, 2048, 25, 42)]
for B, C, O, H, W in shapes:
    x = torch.randn(B, C, H, W, device='cuda').requires_grad_(False); w = (torch.randn(O, C, 1, 1, device='cuda') * 0.05).requires_grad_()
    gy = torch.randn(B, O, H, W, device='cuda')
    def run():
        w.grad = None
        y = c1._ConvSplit.apply(x, w); y.backward(gy)
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    L = c1._lib_sizes()
    nbytes = L.kgdet_conv1x1_grad_weight_workspace_bytes(ctypes.c_int64(B), ctypes.c_int32(O), ctypes.c_int32(C), ctypes.c_int64(H * W))
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda'); gw = torch.empty_like(w)
    from kgdet_amd import _lib
    def k():
        _lib.check(L.kgdet_conv1x1_grad_weight(_lib.ptr(gy), _lib.ptr(x), _lib.ptr(gw), ctypes.c_int64(B), ctypes.c_int32(O), ctypes.c_int32(C),
                                               ctypes.c_int64(H * W), _lib.ptr(ws), ctypes.c_size_t(nbytes), _lib.current_stream()), 'gw')
    for _ in range(5): k()
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): k()
    e1.record(); torch.cuda.synchronize()
    print('C=%4d O=%4d %3dx%-3d  wgrad %6.1f us  (ws %.1f MB)' % (C, O, H, W, e0.elapsed_time(e1) / 50 * 1e3, nbytes / 1e6))
