"""event-timed DeformConv forward on config 5's large maps (3x3, 256 channels): python tools/time_large_fwd.py [prec]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import dcn
prec = sys.argv[1] if len(sys.argv) > 1 else 'split'
dev = torch.device('cuda:0')
torch.manual_seed(0)
for (H, W) in ((50, 84), (100, 168)):
    x = torch.randn(2, 256, H, W, device=dev)
    off = torch.randn(2, 18, H, W, device=dev) * 2
    w = torch.randn(256, 256, 3, 3, device=dev) * 0.01
    with torch.no_grad(), dcn.forward_precision(prec):
        for _ in range(5):
            y = dcn.deform_conv(x, off, w, 1, 1, 1)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                dcn.deform_conv(x, off, w, 1, 1, 1)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    flops = 2.0 * 256 * 256 * 9 * 2 * H * W
    t = sorted(ts)[2]
    print('[2,256,%d,%d] 3x3 %s: %.1f us  %.1f TFLOP/s (incl. weight pack per call)' % (H, W, prec, t, flops / t / 1e6))
