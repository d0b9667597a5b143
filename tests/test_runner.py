"""kgdet_amd.runner: LR schedule known answers for the demo config, checkpoint / resume
equivalence, and the demo-set plumbing end to end on the CPU (dataset -> collate -> detector -> losses -> step)."""
import os

import numpy as np
import pytest
import torch

from kgdet_amd import datasets as ds
from kgdet_amd import runner as rn

HERE = os.path.dirname(os.path.abspath(__file__))
ANN = os.path.join(HERE, 'golden', 'demo_dataset-32.json')
DEMO_LR = dict(policy='step', warmup='linear', warmup_iters=500, warmup_ratio=1.0 / 3, step=[8, 11])


def _lrs(lr_config, base_lr, iters_per_epoch, epochs):
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=base_lr)
    s = rn.LrSchedule(**lr_config)
    s.before_run(opt)
    out, it = [], 0
    for e in range(epochs):
        s.before_train_epoch(opt, e)
        for _ in range(iters_per_epoch):
            s.before_train_iter(opt, it)
            out.append(opt.param_groups[0]['lr'])
            it += 1
    return np.array(out)


def test_demo_lr_schedule_known_answers():
    """configs/kgdet_moment_r50_fpn_1x-demo.py:130-139: Adam 1e-4, linear warm-up 500 it from 1/3, /10 at epochs 8, 11."""
    lr = _lrs(DEMO_LR, 1e-4, iters_per_epoch=100, epochs=12)
    assert np.isclose(lr[0], 1e-4 / 3) and np.isclose(lr[250], 1e-4 * (1 - 0.5 * (2 / 3))) and np.isclose(lr[499], 1e-4 * (1 - (1 / 500) * (2 / 3)))
    assert np.all(np.diff(lr[:501]) > 0) and lr[500] == 1e-4 and np.all(lr[500:800] == 1e-4)
    assert np.allclose(lr[800:1100], 1e-5) and np.allclose(lr[1100:], 1e-6)


def test_lr_schedule_variants():
    # warm-up shorter than an epoch boundary that falls inside it: the epoch's regular rate is what gets scaled
    lr = _lrs(dict(policy='step', warmup='constant', warmup_iters=30, warmup_ratio=0.1, step=[1]), 1.0, 20, 3)
    assert np.allclose(lr[:20], 0.1) and np.allclose(lr[20:30], 0.01) and np.allclose(lr[30:40], 0.1) and np.allclose(lr[40:], 0.1)
    lr = _lrs(dict(policy='step', warmup='exp', warmup_iters=10, warmup_ratio=0.01, step=2, gamma=0.5), 1.0, 10, 5)
    assert np.isclose(lr[0], 0.01) and np.isclose(lr[5], 0.01 ** 0.5) and lr[10] == 1.0 and lr[20] == 0.5 and lr[40] == 0.25
    lr = _lrs(dict(policy='step', step=[3, 5], by_epoch=False, warmup='linear', warmup_iters=2, warmup_ratio=0.5), 1.0, 4, 2)
    assert np.allclose(lr, [0.5, 0.75, 1.0, 0.1, 0.1, 0.01, 0.01, 0.01])
    with pytest.raises(NotImplementedError):
        rn.LrSchedule(policy='cosine')
    with pytest.raises(ValueError):
        rn.LrSchedule(policy='step', step=[1], warmup='bogus', warmup_iters=1)


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.fc = torch.nn.Linear(4, 3)
        self.bn1 = torch.nn.BatchNorm1d(3)

    def forward(self, img, target):
        out = self.bn1(self.fc(img))
        return dict(loss_a=[(out - target).square().mean()], loss_b=out.abs().mean() * 0.1, acc=out.detach().mean())


def _toy_loader(n=6):
    g = torch.Generator().manual_seed(0)
    return [dict(img=torch.randn(5, 4, generator=g), target=torch.randn(5, 3, generator=g)) for _ in range(n)]


def _toy_runner(tmp, lr_config=None):
    torch.manual_seed(0)
    model = _Toy()
    opt = rn.build_optimizer(model, dict(type='Adam', lr=1e-2))
    return rn.Runner(model, opt, work_dir=tmp, lr_config=lr_config or dict(policy='step', step=[2], warmup='linear',
                                                                            warmup_iters=4, warmup_ratio=0.25),
                     optimizer_config=dict(grad_clip=dict(max_norm=35, norm_type=2)), checkpoint_config=dict(interval=1),
                     log_interval=3, logger=lambda s: None)


def test_parse_losses_and_paramwise_optimizer():
    out = _Toy()(**_toy_loader(1)[0])
    loss, log = rn.parse_losses(out)
    assert list(log) == ['loss_a', 'loss_b', 'acc', 'loss'] and torch.isclose(loss, log['loss_a'] + log['loss_b'])
    with pytest.raises(TypeError):
        rn.parse_losses(dict(loss_x=1.0))
    opt = rn.build_optimizer(_Toy(), dict(type='SGD', lr=0.1, momentum=0.9, weight_decay=1e-4,
                                          paramwise_options=dict(bias_lr_mult=2., bias_decay_mult=0., norm_decay_mult=0.)))
    by_shape = {tuple(g['params'][0].shape): g for g in opt.param_groups}
    assert len(opt.param_groups) == 4 and by_shape[(3, 4)]['lr'] == 0.1 and by_shape[(3, 4)]['weight_decay'] == 1e-4
    fc_bias, bn_groups = opt.param_groups[1], opt.param_groups[2:]
    assert fc_bias['lr'] == 0.2 and fc_bias['weight_decay'] == 0.0 and all(g['weight_decay'] == 0.0 and g['lr'] == 0.1 for g in bn_groups)


def test_checkpoint_resume_continues_bit_identically(tmp_path):
    loader = _toy_loader()
    full = _toy_runner(str(tmp_path / 'full')).run(loader, max_epochs=3)
    assert sorted(os.listdir(str(tmp_path / 'full'))) == ['epoch_1.pth', 'epoch_2.pth', 'epoch_3.pth', 'latest.pth']
    ckpt = torch.load(str(tmp_path / 'full' / 'epoch_2.pth'), weights_only=False)
    assert ckpt['meta']['epoch'] == 2 and ckpt['meta']['iter'] == 12 and 'optimizer' in ckpt and 'time' in ckpt['meta']
    part = _toy_runner(str(tmp_path / 'part'))
    with torch.no_grad():
        for p in part.model.parameters():
            p.add_(1.0)                                 # start from different weights: everything must come from the file
    part.resume(str(tmp_path / 'full' / 'epoch_2.pth'))
    assert part.epoch == 2 and part.iter == 12
    part.run(loader, max_epochs=3)
    for a, b in zip(full.model.state_dict().values(), part.model.state_dict().values()):
        assert torch.equal(a, b)
    assert full.current_lr() == part.current_lr() == [1e-3]
    assert len(full.log_history) == 6 and full.log_history[0]['iter'] == 3 and 'loss_a' in full.log_history[0]
    assert np.isclose(full.log_history[0]['lr'], 1e-2 * (1 - (1 - 2 / 4) * 0.75))


def test_demo_set_plumbing_end_to_end_on_cpu(tmp_path):
    """BASELINE config 1 in miniature: demo annotations, synthetic pixels, small scale, tiny detector, 2 iterations."""
    from tests import cpu_ops
    from kgdet_amd import build_detector, configs
    norm = dict(mean=[154.992, 146.197, 140.744], std=[62.757, 64.507, 62.076], to_rgb=True)
    data = ds.DeepFashion2Dataset(ann_file=ANN, img_prefix='/nonexistent/', img_scale=(333, 200), img_norm_cfg=norm,
                                  size_divisor=32, flip_ratio=0.5, with_keypoint=True, with_crowd=False, with_label=True)
    data.load_image = lambda idx: np.random.default_rng(idx).integers(
        0, 256, (data.img_infos[idx]['height'], data.img_infos[idx]['width'], 3), dtype=np.uint8)
    np.random.seed(0)
    sampler = ds.GroupSampler(data, samples_per_gpu=2)
    order = list(sampler)[:4]
    loader = [ds.collate([data[i] for i in order[k:k + 2]]) for k in (0, 2)]
    for b in loader:
        b['img_metas'] = b.pop('img_meta')
    cfg = configs.kgdet_r50_fpn()
    cfg.model['backbone'].update(depth=18)
    cfg.model['neck'].update(in_channels=[64, 128, 256, 512], out_channels=32)
    cfg.model['bbox_head'].update(in_channels=32, feat_channels=32, point_feat_channels=32)
    torch.manual_seed(0)
    model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)

    def process(model, batch, train_mode=True):
        losses = model.forward_train(batch['img'], batch['img_metas'], batch['gt_bboxes'], batch['gt_labels'],
                                     batch['gt_keypoints'])
        loss, log = rn.parse_losses(losses)
        return dict(loss=loss, log_vars=log, num_samples=len(batch['img']))

    r = rn.Runner(model, rn.build_optimizer(model, dict(type='Adam', lr=1e-4)), work_dir=str(tmp_path),
                  lr_config=DEMO_LR, optimizer_config=dict(grad_clip=dict(max_norm=35, norm_type=2)),
                  checkpoint_config=dict(interval=1), log_interval=1, logger=lambda s: None, batch_processor=process)
    before = [p.detach().clone() for p in model.bbox_head.parameters()]
    with cpu_ops.patched():
        r.run(loader, max_epochs=1)
    assert r.iter == 2 and os.path.isfile(str(tmp_path / 'epoch_1.pth'))
    assert all(np.isfinite(rec['loss']) for rec in r.log_history) and len(r.log_history[0]) >= 4 + 10
    assert any(not torch.equal(a, b) for a, b in zip(before, model.bbox_head.parameters()))
    assert np.isclose(r.log_history[1]['lr'], 1e-4 * (1 - (1 - 1 / 500) * (2 / 3)))
