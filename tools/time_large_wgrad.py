"""event-timed DeformConv grad_weight on config 5's large maps (3x3, 256 channels), split-operand gather kernel against the
fp32 kernel (KGDET_OPT_EXACT_BACKWARD), and their agreement: python tools/time_large_wgrad.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import dcn, _lib
dev = torch.device('cuda:0')
torch.manual_seed(0)
needs = dict(input=False, offset=False, mask=False, weight=True, bias=False)
for (H, W) in ((50, 84), (100, 168)):
    x = torch.randn(2, 256, H, W, device=dev)
    off = torch.randn(2, 18, H, W, device=dev) * 2
    w = torch.randn(256, 256, 3, 3, device=dev) * 0.01
    go = torch.randn(2, 256, H, W, device=dev)
    shape = dcn._shape(x, w, (1, 1), (1, 1), (1, 1), 1, 1)
    res = {}
    for exact in (1, 0):
        _lib.check(_lib.lib().kgdet_set_option(0, exact), 'kgdet_set_option')
        for _ in range(3):
            gw = dcn._backward(x, off, None, w, None, go, shape, None, needs)[3]
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                dcn._backward(x, off, None, w, None, go, shape, None, needs)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        res[exact] = (sorted(ts)[2], gw.double())
    _lib.check(_lib.lib().kgdet_set_option(0, 0), 'kgdet_set_option')
    err = (res[0][1] - res[1][1]).abs().max().item() / res[1][1].abs().max().item()
    print('[2,256,%d,%d] 3x3 grad_weight: fp32 kernel %.1f us, split gather %.1f us, max diff / max %.2e' %
          (H, W, res[1][0], res[0][0], err), flush=True)
