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


def _graph_case(config):
    from kgdet_amd import build_detector, configs, synthetic
    from kgdet_amd.dist import DistOptimizerHook

    def make():
        cfg = configs.kgdet_r50_fpn() if config == 'kgdet' else configs.reppoints_kp_r50_fpn(soft_nms=True)
        torch.manual_seed(0)
        model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
        params = [p for p in model.parameters() if p.requires_grad]
        opt = torch.optim.Adam(params, lr=1e-5) if config == 'kgdet' else \
            torch.optim.SGD(params, lr=5e-3, momentum=0.9, weight_decay=1e-4, fused=True)
        return model, opt, DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2))
    return make, synthetic.make_batch(2, 'cuda', seed=0)


@pytest.mark.parametrize('config', ['kgdet', 'serial'])
def test_graphed_train_step_follows_the_eager_steps(config):
    """runner.GraphedTrainStep (forward, losses, backward, clip, optimizer as ONE replayed HIP graph; Adam through the fused step
    with its schedule in device memory, config 5's SGD through torch's fused kernel with the rate in a device scalar): after the
    same number of steps from the same state the parameters equal the eager loop's to rounding (the library GEMMs of the head's
    1x1 output convolutions use float atomics: two EAGER runs differ by as much, tools/determinism_probe.py), the learning rate
    set between replays takes effect, and the optimizer's own step counters are right after sync_optimizer_state()."""
    make, batch = _graph_case(config)
    warm, replays = 3, 4
    m1, o1, h1 = make()
    g = rn.GraphedTrainStep(m1, o1, h1, batch, warmup=warm)
    for k in range(replays):
        if k == 2:
            o1.param_groups[0]['lr'] *= 0.5          # a scheduler's change between two replays
        out = g.step()
    torch.cuda.synchronize()
    assert torch.isfinite(out['loss']).item()
    g.sync_optimizer_state()
    m2, o2, h2 = make()
    for k in range(warm + 1 + replays):             # (the class runs `warm` steps + one in its captured form before the capture)
        if k == warm + 1 + 2:
            o2.param_groups[0]['lr'] *= 0.5
        o = rn.batch_processor(m2, batch)
        h2.step(m2, o2, o['loss'])
    torch.cuda.synchronize()
    if h2._fused is not None and h2._fused._pending:
        h2._fused.sync_optimizer_state(o2)
    # the two runs' UPDATES (parameters minus the common initial state) point the same way and have the same length -- element by
    # element Adam turns a gradient that is rounding noise around zero into a step of +-lr, so single elements may differ by a
    # whole step; a missed learning-rate change (the last two of eight steps at half the rate) would change the length by > 5 %
    m3, _, _ = make()
    u1 = torch.cat([(a.detach() - c.detach()).flatten() for a, c in zip(m1.parameters(), m3.parameters()) if a.requires_grad])
    u2 = torch.cat([(b.detach() - c.detach()).flatten() for b, c in zip(m2.parameters(), m3.parameters()) if b.requires_grad])
    cos = float(torch.dot(u1, u2) / (u1.norm() * u2.norm()))
    ratio = float(u1.norm() / u2.norm())
    assert float(u1.norm()) > 0 and cos >= 0.995 and abs(ratio - 1.0) <= 0.01, (cos, ratio)
    if config == 'kgdet':
        s1 = float(next(iter(o1.state.values()))['step'])
        assert s1 == warm + 1 + replays, s1


def test_graphed_train_step_captures_a_batch_of_mixed_shapes():
    """Round 6: a batch whose images have different pad_shapes (invalid grid points for the smaller one) takes the same
    sync-free step -- valid extents in the fused loss kernels -- so the WHOLE step captures as one HIP graph and its replays
    reproduce the eager step's losses (before: such a batch fell to the host-syncing target path, which cannot be captured)."""
    from kgdet_amd import synthetic
    make, _ = _graph_case('kgdet')
    batch = synthetic.make_batch(2, 'cuda', seed=0, mixed_shapes=True)
    assert batch['img_meta'][1]['pad_shape'] != batch['img_meta'][0]['pad_shape']
    m1, o1, h1 = make()
    g = rn.GraphedTrainStep(m1, o1, h1, batch, warmup=2)
    out = g.step()
    torch.cuda.synchronize()
    assert torch.isfinite(out['loss']).item()
    m2, o2, h2 = make()
    losses = []
    for _ in range(2 + 1 + 1):
        o = rn.batch_processor(m2, batch)
        losses.append(float(o['loss']))
        h2.step(m2, o2, o['loss'])
    torch.cuda.synchronize()
    assert abs(float(out['loss']) - losses[-1]) <= 2e-3 * abs(losses[-1]), (float(out['loss']), losses)


def test_eval_after_graph_replays_sees_the_stepped_weights():
    """ADVICE (round 5): a replayed training graph steps the weights through raw pointers -- no version counter moves -- and the
    autocast cast cache / folded backbone / packed deformable operands are keyed on those counters.  replay -> eval under autocast ->
    replay -> eval must give what a FRESH model holding the same weights gives (the detector's train() / eval() drop the caches)."""
    from kgdet_amd import build_detector, configs, synthetic
    make, batch = _graph_case('kgdet')
    m1, o1, h1 = make()
    o1.param_groups[0]['lr'] = 1e-3          # (steps large enough to change detections' scores)
    g = rn.GraphedTrainStep(m1, o1, h1, batch, warmup=2)

    def evaluate(model):
        model.eval()
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
            feats = model.extract_feat(batch['img'])
            outs = model.bbox_head(feats, batch['img_meta'])
        model.train()
        return [o[0].float().clone() for o in outs]

    g.step()
    first = evaluate(m1)
    for _ in range(3):
        g.step()
    torch.cuda.synchronize()
    second = evaluate(m1)
    cfg = configs.kgdet_r50_fpn()
    fresh = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda()
    fresh.load_state_dict(m1.state_dict())
    want = evaluate(fresh)
    # (bf16 autocast; two model objects may settle on different per-shape kernels, so single elements differ by a bf16 rounding --
    # stale casts of the weights from three Adam steps earlier move EVERY output, by far more)
    for a, b, c in zip(first, second, want):
        stale, fresh_dev = float((a - c).abs().mean()), float((b - c).abs().mean())
        assert stale > 0, 'three more steps must change the outputs'
        assert fresh_dev <= 0.05 * stale, (fresh_dev, stale)
        torch.testing.assert_close(b, c, rtol=1.6e-2, atol=1.6e-2 * float(c.abs().max()))


@pytest.mark.parametrize('config', ['kgdet', 'serial'])
def test_two_eager_steps_are_bit_identical(config):
    """Two eager forward + backward passes of the full-size detector from the same weights on the same batch give the same bits in
    every gradient (185 tensors of the KGDet config) and the same loss.  Round 6 closed the last gap: the heads' 13- / 588- / 166-channel
    1x1 output convolutions left the vendor GEMMs (split-K with float atomics) for the split MFMA kernels, whose partial sums are
    added in a fixed order like everything else on the path (slab fix-ups, sorted contribution lists, fixed-order reductions)."""
    from kgdet_amd import build_detector, configs, synthetic
    from kgdet_amd.runner import batch_processor
    cfg = configs.kgdet_r50_fpn() if config == 'kgdet' else configs.reppoints_kp_r50_fpn(soft_nms=True)
    torch.manual_seed(0)
    model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
    batch = synthetic.make_batch(2, 'cuda', seed=0)

    def grads():
        model.zero_grad(set_to_none=True)
        out = batch_processor(model, batch)
        out['loss'].backward()
        torch.cuda.synchronize()
        return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}, out['loss'].detach().clone()

    grads()            # (kernel selection, first packs)
    a, la = grads()
    b, lb = grads()
    assert torch.equal(la, lb)
    bad = [n for n in a if not torch.equal(a[n], b[n])]
    assert len(a) > 100 and not bad, 'gradients differ between two identical passes: %s' % bad[:8]
