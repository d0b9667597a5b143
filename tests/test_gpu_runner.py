"""f1 + f3 on the GPU: the demo-set input pipeline (DeepFashion2Dataset train pipeline, GroupSampler, collate) feeding
the training runtime (Runner: LR warm-up, DistOptimizerHook, checkpoint, resume) that drives the full-size HIP detector
-- the flow of tools/train.py:94-100 / mmdet/apis/train.py with the reference's demo config values."""
import os

import numpy as np
import pytest
import torch

from kgdet_amd import datasets as ds
from kgdet_amd import runner as rn
from tests.golden import demo_cases

pytestmark = pytest.mark.gpu
DEMO_LR = dict(policy='step', warmup='linear', warmup_iters=500, warmup_ratio=1.0 / 3, step=[8, 11])


def _loader(n_batches):
    data = demo_cases.demo_dataset(test_mode=False, flip_ratio=0.5, with_label=True, with_crowd=False)
    np.random.seed(0)
    order = list(ds.GroupSampler(data, samples_per_gpu=2))[:2 * n_batches]
    batches = [ds.collate([data[i] for i in order[k:k + 2]]) for k in range(0, 2 * n_batches, 2)]
    for b in batches:
        b['img_metas'] = b.pop('img_meta')
    return batches


def _to_device(batch):
    out = dict(batch)
    out['img'] = batch['img'].cuda()
    for k in ('gt_bboxes', 'gt_labels', 'gt_keypoints'):
        out[k] = [t.cuda() for t in batch[k]]
    return out


def _process(model, batch, train_mode=True):
    losses = model.forward_train(batch['img'], batch['img_metas'], batch['gt_bboxes'], batch['gt_labels'],
                                 batch['gt_keypoints'])
    loss, log = rn.parse_losses(losses)
    return dict(loss=loss, log_vars=log, num_samples=len(batch['img']))


def test_runner_trains_the_hip_detector_on_the_demo_pipeline_and_resumes_bit_identically(tmp_path):
    loader = _loader(3)
    assert loader[0]['img'].shape[1] == 3 and max(loader[0]['img'].shape[2:]) <= 1344

    def make():
        cfg, model = demo_cases.demo_detector()
        model = model.cuda()
        opt = rn.build_optimizer(model, dict(type='Adam', lr=1e-4))
        return model, opt

    def runner(model, opt, work):
        return rn.Runner(model, opt, work_dir=str(work), lr_config=DEMO_LR,
                         optimizer_config=dict(grad_clip=dict(max_norm=35, norm_type=2)),
                         checkpoint_config=dict(interval=1), log_interval=1, logger=lambda s: None,
                         batch_processor=_process)

    model, opt = make()
    before = [p.detach().clone() for p in model.bbox_head.parameters()]
    full = runner(model, opt, tmp_path / 'full').run(loader, max_epochs=2, to_device=_to_device)
    assert full.iter == 6 and os.path.isfile(str(tmp_path / 'full' / 'epoch_2.pth'))
    assert all(np.isfinite(rec['loss']) for rec in full.log_history) and len(full.log_history) == 6
    assert {'loss_cls_3', 'loss_bbox_3', 'loss_kpt_3'} <= set(full.log_history[0])
    assert any(not torch.equal(a, b) for a, b in zip(before, model.bbox_head.parameters()))
    # linear warm-up of the demo config: lr_i = 1e-4 * (1 - (1 - i / 500) * (1 - 1 / 3))
    assert np.isclose(full.log_history[3]['lr'], 1e-4 * (1 - (1 - 3 / 500) * (2 / 3)))

    # resume from the first epoch's checkpoint: the second epoch must land on the same weights.  MIOpen's backward
    # kernels are not all deterministic, so "same" is measured against a repeat of the uninterrupted run.
    model2, opt2 = make()
    part = runner(model2, opt2, tmp_path / 'part')
    part.resume(str(tmp_path / 'full' / 'epoch_1.pth'), map_location='cuda')
    assert part.epoch == 1 and part.iter == 3
    part.run(loader, max_epochs=2, to_device=_to_device)
    model3, opt3 = make()
    again = runner(model3, opt3, tmp_path / 'again').run(loader, max_epochs=2, to_device=_to_device)
    spread = max(float((a - b).abs().max()) for a, b in zip(model.state_dict().values(), model3.state_dict().values())
                 if a.is_floating_point())
    diff = max(float((a - b).abs().max()) for a, b in zip(model.state_dict().values(), model2.state_dict().values())
               if a.is_floating_point())
    # (the spread of ONE repeat is itself a random number -- 1e-6 .. 3e-5 over this round's runs, the more of the step runs on
    #  deterministic kernels the smaller --, and Adam turns last-bit gradient differences of near-zero gradients into updates of
    #  up to lr = 1e-4 per step; a resume that lost the optimizer state differs by >= 1e-4 after its first step)
    assert diff <= max(4 * spread, 5e-5), (diff, spread)
    assert again.iter == part.iter == 6
