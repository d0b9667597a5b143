"""Float64 evaluation of the BENCH WORKLOAD's training step -- the KGDet detector (seed 0) on the synthetic batch
(seed 0) of 2 x 800 x 1344, forward + nine losses + backward -- on the CPU: this repo's host graph with the test-side
CPU formulations of the deformable ops (tests/cpu_ops.py: grid_sample + einsum, autograd), everything in float64.
Writes tests/golden/step_f64_golden.npz: the nine losses, the gradient norm of every module group and every parameter
tensor, and sampled gradient slices.  The GPU test (tests/test_gpu_head.py::test_full_size_training_step_matches_float64)
holds the HIP step -- split-bf16 products -- against it: a ground truth instead of a second approximation.

    python tests/golden/make_step_golden.py          (~1 min on 8 cores)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from kgdet_amd import configs, synthetic  # noqa: E402
from kgdet_amd.registry import build_detector  # noqa: E402
from tests import cpu_ops  # noqa: E402


def main():
    torch.set_num_threads(8)
    cfg = configs.kgdet_r50_fpn()
    torch.manual_seed(0)
    model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    model.train()
    model.double()
    batch = synthetic.make_batch(2, 'cpu', seed=0)
    with cpu_ops.patched():
        # (ground truth stays float32 like the point grid: the targets are copies / integer decisions, and the losses
        #  promote to float64 when they meet the float64 predictions)
        losses = model(batch['img'].double(), batch['img_meta'], return_loss=True, gt_bboxes=batch['gt_bboxes'],
                       gt_labels=batch['gt_labels'], gt_keypoints=batch['gt_keypoints'])
        sum(sum(v) for v in losses.values()).backward()
    out = {}
    for k, v in losses.items():
        out['loss:' + k] = np.float64(sum(float(t) for t in v))
    groups = {}
    for name, p in model.named_parameters():
        if p.grad is None:
            continue
        out['norm:' + name] = np.float64(p.grad.norm())
        key = '.'.join(name.split('.')[:2])
        groups[key] = groups.get(key, 0.0) + float(p.grad.pow(2).sum())
    for k, v in groups.items():
        out['group:' + k] = np.float64(v ** 0.5)
    g = dict(model.named_parameters())
    for name in ('backbone.layer2.0.conv1.weight', 'backbone.layer3.5.conv2.weight', 'backbone.layer4.2.conv3.weight',
                 'neck.lateral_convs.0.conv.weight', 'neck.lateral_convs.3.conv.weight', 'neck.lateral_convs.2.conv.weight', 'bbox_head.kp_rep_block_3.cls_dfmconv_7.weight',
                 'bbox_head.cls_convs.0.conv.weight'):
        if name not in g or g[name].grad is None:       # (FPN2 branches that do not reach the head's level carry no gradient)
            continue
        a = g[name].grad.detach().numpy()
        out['grad:' + name] = a.reshape(a.shape[0], -1)[::max(a.shape[0] // 16, 1), ::7].astype(np.float64)
    print({k: float(v) for k, v in out.items() if k.startswith(('loss:', 'group:'))})
    path = os.path.join(HERE, 'step_f64_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
