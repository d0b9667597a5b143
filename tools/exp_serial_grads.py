import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from kgdet_amd import configs
from tests import ref_checks
from tests.golden import ref_cases

G = ref_checks.load('ref_serial_golden.npz')
cfg = configs.reppoints_kp_r50_fpn()
head = ref_cases.serial_head().cuda().train()
xs_cpu, batch = ref_cases.serial_inputs()
xs, outs, losses = ref_checks.head_outputs_and_losses(head, xs_cpu, batch, cfg.train_cfg, 'cuda')
which = sys.argv[1] if len(sys.argv) > 1 else 'all'
tot = sum(sum(v) for k, v in losses.items() if which == 'all' or k == which)
tot.backward()
for lvl, x in enumerate(xs):
    print('grad:x', lvl, ref_checks.rel(x.grad.cpu().numpy()[:, ::8], G['grad:x%d' % lvl]))
params = dict(head.named_parameters())
for key in G.files:
    if key.startswith('gradnorm:'):
        got, want = float(params[key[9:]].grad.norm()), float(G[key])
        print('%-40s %.6e %.6e  rel %.2e' % (key[9:], got, want, abs(got - want) / max(want, 1e-12)))
