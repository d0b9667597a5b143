"""bandwidth of the in-place bias + ReLU epilogue (csrc/epilogue.hip) on the bf16 channels-last activations of the inference batch"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import backbone
dev = torch.device('cuda:0')
for (N, C, H, W) in ((8, 64, 200, 336), (8, 128, 100, 168), (8, 256, 50, 84), (8, 512, 25, 42)):
    ys = [torch.randn(N, C, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(6)]
    b = torch.randn(C, device=dev)
    for y in ys:
        backbone._epilogue_(y, b, None, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        for y in ys:                    # six tensors in turn: none of them stays in the caches
            backbone._epilogue_(y, b, None, True)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 60 * 1e3
    nbytes = ys[0].numel() * 2 * 2
    print('[%d,%d,%d,%d] bf16 NHWC: %.1f us  %.2f TB/s (read + write)' % (N, C, H, W, t, nbytes / t / 1e6))
