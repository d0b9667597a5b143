"""3x3 bf16 channels-last convolution + bias + ReLU at the backbone's inference shapes (B = 8): MIOpen's fused
`torch.miopen_convolution_relu` against convolution + csrc/epilogue.hip's bias_act pass: python tools/bench_conv_relu.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from kgdet_amd import backbone
dev = torch.device('cuda:0')
torch.backends.cudnn.benchmark = os.environ.get("BENCHMARK", "1") == "1"
shapes = [(8, 64, 200, 336, 1), (8, 128, 200, 336, 2), (8, 128, 100, 168, 1), (8, 256, 100, 168, 2), (8, 256, 50, 84, 1),
          (8, 512, 50, 84, 2), (8, 512, 25, 42, 1)]


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (N, C, H, W, s) in shapes:
    O = C
    x = torch.randn(N, C, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(O, C, 3, 3, device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    b32 = torch.randn(O, device=dev)
    b16 = b32.to(torch.bfloat16)

    def sep():
        y = F.conv2d(x, w, None, s, 1)
        return backbone._epilogue_(y, b32, None, True)

    def fused():
        return torch.miopen_convolution_relu(x, w, b16, [s, s], [1, 1], [1, 1], 1)

    try:
        a = sep().float()
        b = fused().float()
        err = float((a - b).abs().max() / a.abs().max())
        print('C=%d %dx%d s%d: conv+bias_act %.1f us, miopen_convolution_relu %.1f us, rel diff %.2e, fused layout cl=%s' % (
            C, H, W, s, timeit(sep), timeit(fused), err, fused().is_contiguous(memory_format=torch.channels_last)))
    except Exception as e:
        print('C=%d %dx%d s%d: fused failed: %s' % (C, H, W, s, str(e)[:200]))
