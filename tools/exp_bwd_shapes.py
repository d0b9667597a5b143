"""forward / backward of deform_conv on a list of shapes vs the float64 oracle (which backward path breaks where)"""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, '.')
import oracle
from kgdet_amd import dcn

CASES = [(2, 256, 32, 40, 256, 3, 1, 1, 1, 1, 1), (2, 256, 16, 20, 256, 3, 1, 1, 1, 1, 1),
         (2, 256, 33, 33, 256, 3, 1, 1, 1, 1, 1), (2, 256, 34, 34, 256, 3, 1, 1, 1, 1, 1),
         (1, 256, 39, 39, 256, 3, 1, 1, 1, 1, 1), (2, 64, 50, 84, 64, 3, 1, 1, 1, 1, 1),
         (2, 256, 50, 84, 256, 3, 1, 1, 1, 1, 1), (2, 256, 100, 168, 256, 3, 1, 1, 1, 1, 1)]
if len(sys.argv) > 1:
    CASES = [CASES[int(i)] for i in sys.argv[1:]]
for case in CASES:
    N, C, H, W, O, k, s, p, d, g, dg = case
    rng = np.random.default_rng(5)
    x = rng.normal(size=(N, C, H, W)).astype(np.float32)
    off = (rng.normal(size=(N, 2 * k * k, H, W)) * 2.0).astype(np.float32)
    w = (rng.normal(size=(O, C, k, k)) * 0.05).astype(np.float32)
    go = rng.normal(size=(N, O, H, W)).astype(np.float32)
    tx, to, tw = (torch.from_numpy(a).cuda().requires_grad_() for a in (x, off, w))
    out = dcn.deform_conv(tx, to, tw, s, p, d, g, dg)
    out.backward(torch.from_numpy(go).cuda())
    torch.cuda.synchronize()
    t0 = time.time()
    f64 = lambda a: a.astype(np.float64)
    ref_out = oracle.deform_conv_forward(f64(x), f64(off), f64(w), s, p, d, g, dg)
    ref = oracle.deform_conv_backward(f64(x), f64(off), f64(w), f64(go), s, p, d, g, dg)
    err = lambda a, b: float(np.abs(a.astype(np.float64) - b).max()) / max(float(np.abs(b).max()), 1e-6)
    print(case, 'fwd %.2e gi %.2e goff %.2e gw %.2e  (oracle %.1f s)' % (
        err(out.detach().cpu().numpy(), ref_out), err(tx.grad.cpu().numpy(), ref['grad_input']),
        err(to.grad.cpu().numpy(), ref['grad_offset']), err(tw.grad.cpu().numpy(), ref['grad_weight']),
        time.time() - t0), flush=True)
