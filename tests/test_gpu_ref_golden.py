"""GPU suite against the REFERENCE-PINNED fixtures (tests/golden/ref_*_golden.npz: outputs of the reference's own
Python modules, generated in the build container by tests/golden/make_ref_golden.py): the HIP-backed heads at full
width, the sync-free dense target path the training step runs, and the HIP focal-loss kernel."""
import numpy as np
import pytest
import torch

from kgdet_amd import focal_loss, losses
from tests import ref_checks
from tests.golden import ref_cases

pytestmark = pytest.mark.gpu

CASES = ref_cases.target_cases()


@pytest.mark.parametrize('name', [n for n in CASES if n != 'kgdet_empty_gt'])
def test_dense_targets_on_gpu_equal_reference(name):
    """points.point_target_kp_dense on the device (what bench.py's training step runs) == point_target_kp.py:98-169, incl. the
    two batches of mixed pad_shapes whose grid has invalid points (`kgdet_invalid_points`, `pyramid_init_pos3`: round 6)"""
    G = ref_checks.load('ref_targets_golden.npz')
    ref_checks.check_points_and_flags(G, name, CASES[name], 'cuda')
    ref_checks.check_point_target(G, name, ref_checks.run_point_target(CASES[name], 'cuda', dense=True))


@pytest.mark.parametrize('name', [n for n in CASES if n != 'kgdet_empty_gt'])
def test_mirrored_targets_on_gpu_equal_reference(name):
    """the list-returning front-end (point_target_kp) of the same dense rules"""
    G = ref_checks.load('ref_targets_golden.npz')
    ref_checks.check_point_target(G, name, ref_checks.run_point_target(CASES[name], 'cuda', dense=False))


def test_empty_ground_truth_raises_on_gpu():
    for dense in (False, True):
        with pytest.raises(ValueError):
            ref_checks.run_point_target(CASES['kgdet_empty_gt'], 'cuda', dense=dense)


def test_hip_focal_kernel_equals_reference_py_sigmoid_focal_loss():
    """csrc/focal.hip forward + backward on 2100 x 13 against the reference's pure-torch focal (focal_loss.py:10-25)
    evaluated in float64; 1e-5 of the largest element (float32 kernel, as the reference's CUDA kernel)"""
    G = ref_checks.load('ref_targets_golden.npz')
    pred, target, weight = ref_cases.focal_inputs()
    p = pred.cuda().requires_grad_(True)
    el = focal_loss.sigmoid_focal_loss(p, target.cuda(), 2.0, 0.25)
    assert ref_checks.rel(el.detach().cpu().numpy(), G['focal:f64:elementwise']) < 1e-5
    assert ref_checks.rel(el.detach().cpu().numpy(), G['focal:f32:elementwise']) < 1e-5
    p2 = pred.cuda().requires_grad_(True)
    total = losses.sigmoid_focal_loss(p2, target.cuda(), weight.cuda(), gamma=2.0, alpha=0.25, reduction='mean',
                                      avg_factor=6.0)
    total.backward()
    want = float(G['focal:f64:weighted_mean'])
    assert abs(float(total.detach()) - want) < 1e-5 * want
    assert ref_checks.rel(p2.grad.cpu().numpy(), G['focal:f64:grad']) < 1e-5


@pytest.mark.parametrize('precision', ['split', 'exact'])
def test_hip_kgdet_head_equals_reference_head(precision):
    """the HIP-backed KGDet head, full width ([2, 256, 25, 42], 588 keypoint channels, 36 deformable convs), against
    the REFERENCE head module: forward maps 2e-4, losses 2e-4, gradients 1e-3, decoded boxes / keypoints 1e-3
    (north-star bar), NMS selection identical"""
    from kgdet_amd import dcn
    head = ref_cases.kgdet_head().cuda()
    dcn.set_forward_precision(precision)
    try:
        worst = ref_checks.check_kgdet_head(head, 'cuda')
    finally:
        dcn.set_forward_precision('split')
    print(precision, worst)


def test_hip_kgdet_head_flip_forward_equals_reference_head():
    """the HIP-backed head with flip_forward=True against the reference head's flip-fused maps (2e-4) and detections (1e-3,
    NMS selection identical)"""
    head = ref_cases.kgdet_head().cuda()
    head.flip_forward = True
    worst = ref_checks.check_kgdet_head_flip(head, 'cuda')
    print(worst)


def test_hip_serial_head_equals_reference_head():
    """config 5: HIP-backed serial head, five pyramid levels, PointAssigner + MaxIoUAssigner targets, hard and soft
    NMS, against the REFERENCE serial head module"""
    head = ref_cases.serial_head().cuda()
    worst = ref_checks.check_serial_head(head, 'cuda')
    print(worst)


def test_hip_parallel_head_equals_reference_head():
    """config 5's sibling: the parallel head (reppoints on their own conv branch, reppoints_head_kp_parallel.py:153-168,
    314-332) on the HIP ops against the REFERENCE parallel head module (tests/golden/ref_parallel_golden.npz; round-2
    review: the parallel head had only been compared with itself)"""
    head = ref_cases.serial_head(parallel=True).cuda()
    worst = ref_checks.check_serial_head(head, 'cuda', golden='ref_parallel_golden.npz', parallel=True)
    print(worst)


def test_hip_serial_head_large_level_equals_reference_head():
    """the serial head on a 384 x 512 pyramid: the stride-8 level has 48 x 64 = 3072 pixels, beyond the 1344-pixel limit
    of the LDS-plane deformable kernels, so the large-map forward / backward kernels are inside a reference-pinned
    comparison (losses, input gradients over the pyramid, gradient norms, hard- and soft-NMS detections)"""
    head = ref_cases.serial_head().cuda()
    worst = ref_checks.check_serial_head(head, 'cuda', golden='ref_serial_large_golden.npz', size=(384, 512))
    print(worst)
