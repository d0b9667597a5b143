"""one 3x3 grad_weight shape in a loop (for counter passes): python tools/one_wgrad3x3.py C O H W [iters]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import conv1x1 as c1, _lib
C, O, H, W = (int(v) for v in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
B = 2
x = torch.randn(B, C, H, W, device='cuda'); gy = torch.randn(B, O, H, W, device='cuda')
L = c1._lib_sizes()
nbytes = L.kgdet_conv3x3_grad_weight_workspace_bytes(ctypes.c_int64(B), ctypes.c_int32(O), ctypes.c_int32(C), ctypes.c_int32(H), ctypes.c_int32(W))
ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda'); gw = torch.empty(O, C, 3, 3, device='cuda')
for _ in range(iters):
    _lib.check(L.kgdet_conv3x3_grad_weight(_lib.ptr(gy), _lib.ptr(x), _lib.ptr(gw), ctypes.c_int64(B), ctypes.c_int32(O), ctypes.c_int32(C),
                                           ctypes.c_int32(H), ctypes.c_int32(W), _lib.ptr(ws), ctypes.c_size_t(nbytes), _lib.current_stream()), 'gw')
torch.cuda.synchronize()
