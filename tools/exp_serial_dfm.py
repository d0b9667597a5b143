"""capture every deform_conv_cat call of the serial head (inputs + incoming gradient) and replay it against the
float64 grid_sample formulation on the GPU"""
import sys
import torch
sys.path.insert(0, '.')
from kgdet_amd import configs, dcn
from tests import ref_checks, torch_ref
from tests.golden import ref_cases

cfg = configs.reppoints_kp_r50_fpn()
head = ref_cases.serial_head().cuda().train()
xs_cpu, batch = ref_cases.serial_inputs()
calls = []
orig = dcn.deform_conv_cat


def spy(x, offsets, weights, pads, relu=True):
    out = orig(x, offsets, weights, pads, relu)
    rec = dict(x=x.detach().clone(), off=offsets[0].detach().clone(), w=weights[0].detach().clone(), pad=pads[0])
    out.register_hook(lambda g, rec=rec: rec.__setitem__('g', g.detach().clone()))
    calls.append(rec)
    return out


dcn.deform_conv_cat = spy
import kgdet_amd.heads_serial as hs
hs.dcn.deform_conv_cat = spy
xs, outs, losses = ref_checks.head_outputs_and_losses(head, xs_cpu, batch, cfg.train_cfg, 'cuda')
sum(sum(v) for v in losses.values()).backward()
dcn.deform_conv_cat = orig
rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max().clamp(min=1e-30))
for i, r in enumerate(calls):
    if 'g' not in r:
        continue
    x, off, w = (r[k].clone().requires_grad_() for k in ('x', 'off', 'w'))
    y = orig(x, [off], [w], [r['pad']])
    y.backward(r['g'])
    xd, od, wd = (r[k].double().requires_grad_() for k in ('x', 'off', 'w'))
    pre = torch_ref.deform_conv(xd, od, wd, 1, r['pad'], 1)
    flips = int(((pre > 0) != (y > 0)).sum())
    yd = pre * (y > 0).double()          # the ReLU decisions of the HIP forward: isolates the backward arithmetic
    yd.backward(r['g'].double())
    print('   relu decisions that differ from float64:', flips, 'of', pre.numel(), ' min |pre| %.2e' % float(pre.abs().min()))
    print(i, tuple(r['x'].shape), 'off absmax %.3f' % float(r['off'].abs().max()), 'fwd %.2e gi %.2e goff %.2e gw %.2e |g| %.2e' % (
        rel(y, yd), rel(x.grad, xd.grad), rel(off.grad, od.grad), rel(w.grad, wd.grad), float(r['g'].abs().max())), flush=True)
