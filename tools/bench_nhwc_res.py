"""conv3 + bias + residual + ReLU of the bottlenecks at the inference batch (bf16 channels-last, B = 8): the fused kernel
(csrc/conv_nhwc.hip) against MIOpen's convolution + the epilogue pass and against hipBLASLt's GEMM with the residual as C + the
epilogue pass:  python tools/bench_nhwc_res.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from kgdet_amd import backbone, _lib
dev = torch.device('cuda:0')
torch.backends.cudnn.benchmark = True
L = _lib.lib()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (B, K, H, W) in ((8, 64, 200, 336), (8, 128, 100, 168), (8, 256, 50, 84)):
    N = 4 * K
    x = torch.randn(B, K, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    r = torch.randn(B, N, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(N, K, 1, 1, device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    b = torch.randn(N, device=dev)
    out = torch.empty_like(r)

    def conv_path():
        return backbone._epilogue_(F.conv2d(x, w), b, r, True)

    def gemm_path():
        y2 = torch.addmm(r.permute(0, 2, 3, 1).reshape(-1, N), x.permute(0, 2, 3, 1).reshape(-1, K), w.view(N, K).t())
        return backbone._epilogue_(y2.view(B, H, W, N).permute(0, 3, 1, 2), b, None, True)

    def fused():
        _lib.check(L.kgdet_conv1x1_nhwc_residual(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(r), _lib.ptr(out),
                                                 ctypes.c_int64(B * H * W), ctypes.c_int32(K), ctypes.c_int32(N), ctypes.c_int32(1),
                                                 _lib.current_stream()), 'nhwc')
        return out

    a = conv_path().float()
    c = fused().float()
    err = float((a - c).abs().max() / a.abs().max())
    nbytes = (x.numel() + 2 * r.numel()) * 2
    tf = timeit(fused)
    print('K=%d N=%d %dx%d: conv+epilogue %.1f us, gemm+epilogue %.1f us, fused %.1f us (%.2f TB/s), max diff %.1e' % (
        K, N, H, W, timeit(conv_path), timeit(gemm_path), tf, nbytes / tf / 1e6, err))
