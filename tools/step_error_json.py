"""The split-operand training step's measured deviation from the float64 golden of the same step
(tests/golden/step_f64_golden.npz; the comparison of tests/test_gpu_head.py::test_full_size_training_step_matches_float64) as one
JSON object -- bench.py prints it as `value_error_vs_f64` next to `dtype`:
    python tools/step_error_json.py [split|exact] > gpurun_out/step_vs_f64.json      (copy to profiles/r06_step_vs_f64.json)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from kgdet_amd import configs, conv1x1, dcn, synthetic
from kgdet_amd.registry import build_detector
mode = sys.argv[1] if len(sys.argv) > 1 else 'split'
G = np.load(os.path.join(ROOT, 'tests', 'golden', 'step_f64_golden.npz'))
cfg = configs.kgdet_r50_fpn()
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
batch = synthetic.make_batch(2, 'cuda', seed=0)
conv1x1._entries.clear(); conv1x1._fold_entries.clear()
with dcn.arithmetic(mode):
    for rep in range(2):      # (the second pass runs with the BatchNorms folded: the step the bench times)
        model.zero_grad()
        losses = model(batch['img'], batch['img_meta'], return_loss=True, gt_bboxes=batch['gt_bboxes'],
                       gt_labels=batch['gt_labels'], gt_keypoints=batch['gt_keypoints'])
        sum(sum(v) for v in losses.values()).backward()
torch.cuda.synchronize()
loss_dev = {k: abs(sum(float(t) for t in v) - float(G['loss:' + k])) / max(1.0, abs(float(G['loss:' + k]))) for k, v in losses.items()}
groups, params = {}, dict(model.named_parameters())
for name, p in params.items():
    if p.grad is not None:
        key = '.'.join(name.split('.')[:2])
        groups[key] = groups.get(key, 0.0) + float(p.grad.double().pow(2).sum())
dev = {k: abs(v ** 0.5 - float(G['group:' + k])) / float(G['group:' + k]) for k, v in groups.items()}


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max(initial=0)) / max(float(np.abs(b).max(initial=0)), 1e-30)


worst_slice, worst_slice_key = 0.0, None
for key in G.files:
    if key.startswith('grad:'):
        a = params[key[5:]].grad.detach().cpu().numpy()
        a = a.reshape(a.shape[0], -1)[::max(a.shape[0] // 16, 1), ::7]
        r = rel(a, G[key])
        if r > worst_slice:
            worst_slice, worst_slice_key = r, key[5:]
wg = max(dev, key=dev.get)
print(json.dumps({'arithmetic': mode, 'worst_gradient_group_norm_rel_dev': float('%.3g' % dev[wg]), 'worst_gradient_group': wg,
                  'worst_loss_rel_dev': float('%.3g' % max(loss_dev.values())),
                  'worst_sampled_gradient_slice_rel_dev': float('%.3g' % worst_slice), 'worst_sampled_slice': worst_slice_key,
                  'gradient_groups': len(dev), 'golden': 'tests/golden/step_f64_golden.npz (this build\'s graph evaluated in float64 on the CPU)',
                  'workload': 'the bench step: 2 x 800 x 1344, seeds 0, second pass (BatchNorms folded)'}))
