"""event-timed grouped head-stage forward (the bench's roofline launch): python tools/time_group.py [B] [prec] [iters]
prints the median-of-5 time of one launch sequence; KGDET_LIB selects an experiment build of the library"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import dcn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
prec = sys.argv[2] if len(sys.argv) > 2 else 'split'
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device('cuda:0')
torch.manual_seed(0)
C, H, W = 256, 25, 42
xs = [torch.randn(B, C, H, W, device=dev) for _ in range(2)]
ks = (3, 5, 7)
offs = [torch.randn(B, 2 * k * k, H, W, device=dev) * 2 for k in ks]
ws = [[torch.randn(C, C, k, k, device=dev) * 0.01 for k in ks] for _ in xs]
pads = [k // 2 for k in ks]
st = torch.cuda.current_stream()
with torch.no_grad(), dcn.forward_precision(prec):
    for _ in range(5):
        dcn.deform_conv_cat_multi(xs, offs, ws, pads)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(iters):
            dcn.deform_conv_cat_multi(xs, offs, ws, pads)
        e1.record(st)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
flops = sum(2.0 * C * C * k * k * B * H * W for k in ks) * 2
t = sorted(ts)[2]
print('%s B=%d %s: %.1f us per launch sequence (min %.1f)  %.1f TFLOP/s' % (
    os.environ.get('KGDET_LIB', 'product'), B, prec, t, min(ts), flops / t / 1e6))
