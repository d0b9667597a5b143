"""per-parameter gradient norms of the full-size training step on the HIP path against the float64 golden
(tests/golden/step_f64_golden.npz): python tools/step_vs_f64.py [split|exact]  -- signed relative deviations"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from kgdet_amd import configs, dcn, synthetic
from kgdet_amd.registry import build_detector
mode = sys.argv[1] if len(sys.argv) > 1 else 'split'
if os.environ.get('KGDET_EXP_NO_CONVMODULE_SPLIT'):     # experiment: ConvModule convolutions (towers, FPN) on MIOpen fp32
    from kgdet_amd import layers
    _app = layers.conv1x1.applicable
    import types
    layers.conv1x1 = types.SimpleNamespace(applicable=lambda *a, **k: False, conv_split=None)
G = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'step_f64_golden.npz'))
cfg = configs.kgdet_r50_fpn()
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
batch = synthetic.make_batch(2, 'cuda', seed=0)
import contextlib
exp = os.environ.get('KGDET_EXP', '')      # comma list of: dcn_fwd_exact, dcn_bwd_exact, conv_exact
if exp:
    from kgdet_amd import _lib, conv1x1
    if 'dcn_fwd_exact' in exp:
        dcn.set_forward_precision('exact')
    if 'dcn_bwd_exact' in exp:
        dcn._EXACT_BACKWARD = True
        _lib.check(_lib.lib().kgdet_set_option(0, 1), 'opt')
    if 'conv_exact' in exp:
        conv1x1.ENABLED = False
    import types
    if 'backbone_exact' in exp:        # only the backbone's convolutions on MIOpen fp32
        import kgdet_amd.backbone as bb
        bb.conv1x1 = types.SimpleNamespace(**{k: getattr(conv1x1, k) for k in dir(conv1x1) if not k.startswith('__')})
        bb.conv1x1.applicable = lambda *a, **k: False
        bb.conv1x1.applicable_stride2 = lambda *a, **k: False
        bb.STEM_CONV = False
    if 'biasact_exact' in exp:         # only the head's biased stage-1 convolutions
        import kgdet_amd.heads as hh
        hh.conv1x1 = types.SimpleNamespace(conv_bias_act=lambda conv, x, relu=False: torch.relu(conv(x)) if relu else conv(x))
    if 'fwd_exact' in exp:             # every split dense-convolution FORWARD on fp32 (backward stays split)
        import torch.nn.functional as F
        import kgdet_amd.backbone as bb
        bb.STEM_CONV = False
        cur = {}
        orig_fi = conv1x1.forward_images

        orig_ap = conv1x1._apply

        def fi(x, weight):
            res = orig_fi(x, weight)
            cur['w'], cur['img'] = weight, res[0]
            return res

        def ap(img, x, M, taps, stride=1, bias=None, residual=None, relu=False):
            if img is not cur.get('img'):      # grad_input (the transposed image): stays on the split kernel
                return orig_ap(img, x, M, taps, stride, bias, residual, relu)
            w = cur['w']
            assert w.shape[0] == M and w.shape[2] * w.shape[3] == taps
            y = F.conv2d(x, w, bias, stride, w.shape[2] // 2)
            if residual is not None:
                y = y + residual
            return torch.relu(y) if relu else y
        conv1x1.forward_images, conv1x1._apply = fi, ap
    if 'gw_exact' in exp:              # split forward / grad_input, weight gradients through ATen
        def gw(x, weight, gy):
            k = weight.shape[2]
            return torch.ops.aten.convolution_backward(gy, x, weight, None, [1, 1], [k // 2, k // 2], [1, 1], False, [0, 0], 1,
                                                       [False, True, False])[1]
        conv1x1.grad_weight = gw
with (contextlib.nullcontext() if exp else dcn.arithmetic(mode)):
    losses = model(batch['img'], batch['img_meta'], return_loss=True, gt_bboxes=batch['gt_bboxes'],
                   gt_labels=batch['gt_labels'], gt_keypoints=batch['gt_keypoints'])
    sum(sum(v) for v in losses.values()).backward()
rows = []
for name, p in model.named_parameters():
    if p.grad is not None and 'norm:' + name in G.files:
        want = float(G['norm:' + name])
        rows.append(((float(p.grad.double().norm()) - want) / max(want, 1e-30), name, tuple(p.shape)))
rows.sort(key=lambda r: -abs(r[0]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for d, n, s in [r for r in rows if flt in r[1]][:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print('%+.2e  %-50s %s' % (d, n, s))
print('median |dev| %.2e over %d tensors' % (float(np.median([abs(r[0]) for r in rows])), len(rows)))
