import sys
import numpy as np, torch
sys.path.insert(0, '.')
import oracle
from kgdet_amd import dcn
for N in [int(a) for a in sys.argv[1:]] or (64, 128, 192):
    case = (N, 16, 25, 42, 16, 3, 1, 1, 1, 1, 1)
    N, C, H, W, O, k, s, p, d, g, dg = case
    rng = np.random.default_rng(21)
    x = rng.normal(size=(N, C, H, W)).astype(np.float32)
    off = (rng.normal(size=(N, 2 * k * k, H, W)) * 2.0).astype(np.float32)
    w = (rng.normal(size=(O, C, k, k)) * 0.05).astype(np.float32)
    go = rng.normal(size=(N, O, H, W)).astype(np.float32)
    tx, to, tw = (torch.from_numpy(a).cuda().requires_grad_() for a in (x, off, w))
    out = dcn.deform_conv(tx, to, tw, s, p, d, g, dg)
    out.backward(torch.from_numpy(go).cuda())
    torch.cuda.synchronize()
    f64 = lambda a: a.astype(np.float64)
    ref_o = oracle.deform_conv_forward(f64(x), f64(off), f64(w), s, p, d, g, dg)
    ref = oracle.deform_conv_backward(f64(x), f64(off), f64(w), f64(go), s, p, d, g, dg)
    err = lambda a, b: float(np.abs(a.astype(np.float64) - b).max()) / max(float(np.abs(b).max()), 1e-6)
    gi = tx.grad.cpu().numpy()
    e_img = [err(gi[i], ref['grad_input'][i]) for i in range(N)]
    print(N, 'fwd %.2e gi %.2e goff %.2e gw %.2e' % (err(out.detach().cpu().numpy(), ref_o), err(gi, ref['grad_input']),
          err(to.grad.cpu().numpy(), ref['grad_offset']), err(tw.grad.cpu().numpy(), ref['grad_weight'])),
          'bad gi images', [i for i, e in enumerate(e_img) if e > 1e-4][:10],
          'bad goff images', [i for i in range(N) if err(to.grad[i].cpu().numpy(), ref['grad_offset'][i]) > 1e-4][:12], flush=True)
