"""CPU suite: this repo's host logic (points, assigners, targets, losses, head wiring, decode, post-processing)
against fixtures produced by the REFERENCE'S OWN PYTHON (tests/golden/make_ref_golden.py: the reference modules are
imported in place in the build container; only their outputs travel).  The GPU suite repeats the head / target /
focal comparisons with the HIP ops (tests/test_gpu_ref_golden.py)."""
import numpy as np
import pytest
import torch

from kgdet_amd import losses, points
from tests import cpu_ops, ref_checks, torch_ref
from tests.golden import ref_cases


@pytest.fixture(scope='module')
def G():
    return ref_checks.load('ref_targets_golden.npz')


CASES = ref_cases.target_cases()
OK_CASES = [n for n in CASES if n != 'kgdet_empty_gt']
DENSE_CASES = OK_CASES      # (round 6: incl. the two batches of mixed pad_shapes -- invalid grid points)


@pytest.mark.parametrize('name', OK_CASES)
def test_point_generator_and_mirrored_targets_equal_reference(G, name):
    """point_generator.py:4-34 + point_target_kp.py:98-169 + point_assigner.py / max_iou_assigner.py: bit-exact"""
    ref_checks.check_points_and_flags(G, name, CASES[name], 'cpu')
    ref_checks.check_point_target(G, name, ref_checks.run_point_target(CASES[name], 'cpu', dense=False))


@pytest.mark.parametrize('name', DENSE_CASES)
def test_dense_sync_free_targets_equal_reference(G, name):
    """the dense target path the training step actually runs (points.point_target_kp_dense): bit-exact"""
    ref_checks.check_point_target(G, name, ref_checks.run_point_target(CASES[name], 'cpu', dense=True))


def test_empty_ground_truth_raises_like_the_reference(G):
    assert str(G['kgdet_empty_gt:error']) == 'ValueError'
    for dense in (False, True):
        with pytest.raises(ValueError):
            ref_checks.run_point_target(CASES['kgdet_empty_gt'], 'cpu', dense=dense)
    assert str(G['pointassign:no_gt:error']) == 'ValueError'


def test_bbox_overlaps_equal_reference(G):
    a, b = ref_cases.overlap_boxes()
    for mode in ('iou', 'iof'):
        assert np.array_equal(points.bbox_overlaps(a, b, mode=mode).numpy(), G['overlaps:%s' % mode])
        assert np.array_equal(points.bbox_overlaps(a[:7], b, mode=mode, is_aligned=True).numpy(),
                              G['overlaps:%s_aligned' % mode])


@pytest.mark.parametrize('tag', list(ref_cases.max_iou_cases()))
def test_max_iou_assigner_equals_reference(G, tag):
    a, b = ref_cases.overlap_boxes()
    r = points.MaxIoUAssigner(**ref_cases.max_iou_cases()[tag]).assign(a, b, None, ref_cases.overlap_labels())
    assert np.array_equal(r.gt_inds.numpy(), G['maxiou:%s:gt_inds' % tag])
    assert np.array_equal(r.max_overlaps.numpy(), G['maxiou:%s:max_overlaps' % tag])
    assert np.array_equal(r.labels.numpy(), G['maxiou:%s:labels' % tag])


@pytest.mark.parametrize('tag', list(ref_cases.point_assigner_cases()))
def test_point_assigner_equals_reference(G, tag):
    pts, gts, labels, kw = ref_cases.point_assigner_cases()[tag]
    if tag == 'no_gt':
        with pytest.raises(ValueError):
            points.PointAssigner(**kw).assign(pts, gts, None, labels)
        return
    r = points.PointAssigner(**kw).assign(pts, gts, None, labels)
    assert np.array_equal(r.gt_inds.numpy(), G['pointassign:%s:gt_inds' % tag])
    if labels is not None:
        assert np.array_equal(r.labels.numpy(), G['pointassign:%s:labels' % tag])


def test_focal_formula_and_smooth_l1_equal_reference(G):
    """the test-side focal formula (what the oracle and the HIP kernel are held to) against the reference's own
    py_sigmoid_focal_loss (focal_loss.py:10-25); SmoothL1Loss against smooth_l1_loss.py"""
    pred, target, weight = ref_cases.focal_inputs()
    el = torch_ref.py_sigmoid_focal_loss(pred.double(), target, 2.0, 0.25)
    assert ref_checks.rel(el.numpy(), G['focal:f64:elementwise']) < 1e-12
    el32 = torch_ref.py_sigmoid_focal_loss(pred, target, 2.0, 0.25)
    assert ref_checks.rel(el32.numpy(), G['focal:f64:elementwise']) < 1e-5
    with cpu_ops.patched():
        p = pred.clone().requires_grad_(True)
        total = losses.sigmoid_focal_loss(p, target, weight, gamma=2.0, alpha=0.25, reduction='mean', avg_factor=6.0)
        total.backward()
    assert abs(float(total) - float(G['focal:f64:weighted_mean'])) < 1e-5 * float(G['focal:f64:weighted_mean'])
    assert ref_checks.rel(p.grad.numpy(), G['focal:f64:grad']) < 1e-5
    sp, st, sw = ref_cases.smooth_l1_inputs()
    assert np.array_equal(losses.smooth_l1_loss(sp, st, beta=0.11, reduction='none').numpy(), G['smooth_l1:elementwise'])
    got = losses.SmoothL1Loss(beta=0.11, loss_weight=0.5)(sp, st, sw, avg_factor=7.0)
    assert abs(float(got) - float(G['smooth_l1:weighted'])) < 1e-6 * float(G['smooth_l1:weighted'])


def test_kgdet_head_host_logic_equals_reference_head_on_cpu():
    """this repo's KGDet head (wiring, gradient_mul trick, offsets, moment boxes, targets, nine losses, decode,
    multiclass NMS) with the test-side CPU ops in place of the HIP ops == the reference's head module"""
    head = ref_cases.kgdet_head()
    with cpu_ops.patched():
        worst = ref_checks.check_kgdet_head(head, 'cpu')
    print(worst)


def test_kgdet_head_flip_forward_equals_reference_head_on_cpu():
    """flip_forward=True (KP3:448-488, off in every shipped config): the nine flip-fused maps and the detections of this repo's
    head == the reference head's, with the dataset's real left / right keypoint permutation"""
    head = ref_cases.kgdet_head()
    head.flip_forward = True
    with cpu_ops.patched():
        worst = ref_checks.check_kgdet_head_flip(head, 'cpu')
    print(worst)


def test_serial_head_host_logic_equals_reference_head_on_cpu():
    head = ref_cases.serial_head()
    with cpu_ops.patched():
        worst = ref_checks.check_serial_head(head, 'cpu')
    print(worst)


def test_parallel_head_host_logic_equals_reference_head_on_cpu():
    """RepPointsHeadKpParallel (host logic, test-side CPU ops) == the reference's reppoints_head_kp_parallel.py module"""
    head = ref_cases.serial_head(parallel=True)
    with cpu_ops.patched():
        worst = ref_checks.check_serial_head(head, 'cpu', golden='ref_parallel_golden.npz', parallel=True)
    print(worst)
