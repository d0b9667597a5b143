"""inference 1x1 convolutions in bf16 channels-last: MIOpen conv + the in-place bias/ReLU epilogue (csrc/epilogue.hip) against
one hipBLASLt GEMM with the bias + ReLU epilogue (torch._addmm_activation on the [B*H*W, Cin] view)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from kgdet_amd import backbone as bb
torch.backends.cudnn.benchmark = True
B = 8
shapes = [(64, 64, 200, 336), (64, 256, 200, 336), (256, 64, 200, 336), (256, 128, 200, 336), (128, 512, 100, 168), (512, 128, 100, 168),
          (512, 256, 100, 168), (256, 1024, 50, 84), (1024, 256, 50, 84), (1024, 512, 50, 84), (512, 2048, 25, 42), (2048, 512, 25, 42)]


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot_a = tot_b = 0.0
for cin, cout, H, W in shapes:
    x = torch.randn(B, cin, H, W, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 1, 1, device='cuda') * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    bias = torch.randn(cout, device='cuda')
    bias16 = bias.to(torch.bfloat16)
    w2 = w.view(cout, cin)

    def a():
        y = F.conv2d(x, w)
        return bb._epilogue_(y, bias, None, True)

    def b():
        x2 = x.permute(0, 2, 3, 1).reshape(-1, cin)
        y2 = torch._addmm_activation(bias16, x2, w2.t(), use_gelu=False)
        return y2.view(B, H, W, cout).permute(0, 3, 1, 2)
    ya, yb = a(), b()
    err = float((ya.float() - yb.float()).abs().max() / ya.float().abs().max())
    ta, tb = timeit(a), timeit(b)
    tot_a += ta
    tot_b += tb
    print('%5d -> %5d @ %3dx%3d  conv+epilogue %7.1f us   gemm+fused epilogue %7.1f us   rel diff %.1e  channels_last out: %s' % (
        cin, cout, H, W, ta, tb, err, yb.is_contiguous(memory_format=torch.channels_last)))
print('total %.1f vs %.1f us' % (tot_a, tot_b))
