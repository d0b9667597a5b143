"""1x1 convolution (stride 1, fp32 NCHW) on the bf16 hi/lo-split MFMA GEMM kernels (csrc/conv1x1.hip).

``conv1x1(x, weight)`` equals ``F.conv2d(x, weight)`` for a ``[O, C, 1, 1]`` weight to fp32 accuracy (~1e-6 of the
output scale), forward and both gradients; used by the backbone's bottlenecks (kgdet_amd/backbone.py) where MIOpen's
fp32 GEMMs run at 60-110 TFLOP/s."""
import ctypes

import torch

from . import _lib


def applicable(x, weight, stride=(1, 1), padding=(0, 0), dilation=(1, 1), groups=1):
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4
            and weight.shape[2:] == (1, 1) and tuple(stride) == (1, 1) and tuple(padding) == (0, 0) and groups == 1
            and x.is_contiguous() and weight.shape[1] % 16 == 0 and weight.shape[0] % 16 == 0
            and (x.shape[2] * x.shape[3]) % 2 == 0 and not torch.is_autocast_enabled())


def _lib_sizes():
    L = _lib.lib()
    L.kgdet_conv1x1_packed_bytes.restype = ctypes.c_size_t
    L.kgdet_conv1x1_grad_weight_workspace_bytes.restype = ctypes.c_size_t
    L.kgdet_conv1x1_apply_workspace_bytes.restype = ctypes.c_size_t
    return L


def _pack(weight2d, transpose):
    L = _lib_sizes()
    O, C = weight2d.shape
    M, K = (C, O) if transpose else (O, C)
    nbytes = L.kgdet_conv1x1_packed_bytes(ctypes.c_int32(M), ctypes.c_int32(K))
    img = torch.empty(nbytes, dtype=torch.uint8, device=weight2d.device)
    _lib.check(L.kgdet_conv1x1_pack(_lib.ptr(weight2d), ctypes.c_int32(O), ctypes.c_int32(C),
                                    ctypes.c_int32(1 if transpose else 0), _lib.ptr(img), _lib.current_stream()),
               'conv1x1_pack')
    return img


def _apply(img, x, M):
    L = _lib_sizes()
    B, K, H, W = x.shape
    y = torch.empty((B, M, H, W), dtype=torch.float32, device=x.device)
    nbytes = L.kgdet_conv1x1_apply_workspace_bytes(ctypes.c_int64(B), ctypes.c_int32(M), ctypes.c_int32(K),
                                                   ctypes.c_int64(H * W))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
    _lib.check(L.kgdet_conv1x1_apply(_lib.ptr(img), _lib.ptr(x), _lib.ptr(y), ctypes.c_int64(B), ctypes.c_int32(M),
                                     ctypes.c_int32(K), ctypes.c_int64(H * W), _lib.ptr(ws), ctypes.c_size_t(nbytes),
                                     _lib.current_stream()), 'conv1x1_apply')
    return y


class _Conv1x1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        w2 = weight.reshape(weight.shape[0], weight.shape[1])
        ctx.save_for_backward(x, weight)
        return _apply(_pack(w2, False), x, w2.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        O, C = weight.shape[0], weight.shape[1]
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _apply(_pack(weight.reshape(O, C), True), gy, C)
        if ctx.needs_input_grad[1] and (x.shape[2] * x.shape[3]) % 4 != 0:
            # 8-byte-load variant of the kernel (small odd maps, e.g. 25 x 42): MIOpen's fp32 GEMM is faster there
            gw = torch.nn.grad.conv2d_weight(x, weight.shape, gy)
        elif ctx.needs_input_grad[1]:
            L = _lib_sizes()
            B, HW = x.shape[0], x.shape[2] * x.shape[3]
            nbytes = L.kgdet_conv1x1_grad_weight_workspace_bytes(ctypes.c_int64(B), ctypes.c_int32(O),
                                                                 ctypes.c_int32(C), ctypes.c_int64(HW))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            gw = torch.empty_like(weight)
            _lib.check(L.kgdet_conv1x1_grad_weight(_lib.ptr(gy), _lib.ptr(x), _lib.ptr(gw), ctypes.c_int64(B),
                                                   ctypes.c_int32(O), ctypes.c_int32(C), ctypes.c_int64(HW),
                                                   _lib.ptr(ws), ctypes.c_size_t(nbytes), _lib.current_stream()),
                       'conv1x1_grad_weight')
        return gx, gw


def conv1x1(x, weight):
    return _Conv1x1.apply(x, weight)
