"""Random-shape stress of the split-operand plane kernels (forward, grad_input, grad_offset, grad_weight; v1 and v2) against
the exact-fp32 kernels: python tools/stress_plane.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kgdet_amd import dcn
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device('cuda:0')
worst = 0.0
for case in range(n_cases):
    N = int(rng.integers(1, 9))
    C = int(rng.choice([16, 32, 48, 64, 96, 128, 256]))
    O = int(rng.choice([16, 32, 64, 128, 256, 512]))
    kh, kw = [(1, 1), (3, 3), (3, 3), (5, 5), (7, 7), (3, 1), (1, 3)][int(rng.integers(0, 7))]
    H, W = int(rng.integers(4, 33)), int(rng.integers(4, 41))
    v2 = bool(rng.integers(0, 2))
    sigma = float(rng.choice([0.3, 1.0, 3.0]))
    g = torch.Generator(device='cpu').manual_seed(case)
    x = torch.randn(N, C, H, W, generator=g).to(dev)
    off = (torch.randn(N, 2 * kh * kw, H, W, generator=g) * sigma).to(dev)
    w = (torch.randn(O, C, kh, kw, generator=g) * 0.05).to(dev)
    m = torch.rand(N, kh * kw, H, W, generator=g).to(dev) if v2 else None
    go = torch.randn(N, O, H, W, generator=g).to(dev)
    pad = (kh // 2, kw // 2)
    res = {}
    for mode in ('split', 'exact'):
        xs, os_, ws = x.clone().requires_grad_(), off.clone().requires_grad_(), w.clone().requires_grad_()
        ms = m.clone().requires_grad_() if v2 else None
        with dcn.arithmetic(mode):
            if v2:
                out = dcn.modulated_deform_conv(xs, os_, ms, ws, None, 1, pad, 1, 1, 1)
            else:
                out = dcn.deform_conv(xs, os_, ws, 1, pad, 1, 1, 1)
            out.backward(go)
        res[mode] = [out.detach(), xs.grad, os_.grad, ws.grad] + ([ms.grad] if v2 else [])
    errs = []
    for a, b in zip(res['split'], res['exact']):
        errs.append(float((a - b).abs().max() / b.abs().max().clamp_min(1e-20)))
    worst = max(worst, max(errs))
    flag = '' if max(errs) < 1e-4 and all(torch.isfinite(t).all() for t in res['split']) else '   <-- CHECK'
    print('case %2d N=%d C=%3d O=%3d k=%dx%d %2dx%2d v2=%d sigma=%.1f  rel err fwd/gx/goff/gw%s: %s%s' % (
        case, N, C, O, kh, kw, H, W, v2, sigma, '/gm' if v2 else '', ' '.join('%.1e' % e for e in errs), flag))
print('worst', worst)
