"""BASELINE config 1 / north star "bbox / keypoint AP on the demo set matching the reference within +-0.1", on the
GPU: all 32 demo images (rendered from the reference-held annotation file) -> DeepFashion2Dataset test pipeline ->
RepPointsDetectorKp on the HIP kernels -> results2json -> CocoEvaluator, against the REFERENCE detector's own
detections and COCOeval statistics from its CPU path (tests/golden/make_demo_golden.py)."""
import pytest

from tests import demo_checks
from tests.golden import demo_cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('precision', ['split', 'exact'])
def test_demo_set_detections_and_ap_match_reference_cpu_path(tmp_path, precision):
    from kgdet_amd import dcn
    cfg, model = demo_cases.demo_detector()
    model = model.cuda()
    prev = dcn.set_forward_precision(precision)
    try:
        out = demo_checks.check_demo_set(model, 'cuda', list(range(32)), tmp_path, full_set=True)
    finally:
        dcn.set_forward_precision(prev)
    print(precision, out)
    assert out['detections'] == 401


def test_test_time_harness_on_the_hip_detector(tmp_path):
    """runner.multi_gpu_test (tools/test.py:38-100) on the GPU in a ONE-rank nccl group -- what a 1-GPU box can form: the sharded,
    gathered results are the per-image results of the reference-style loop (demo_checks.run_detector), for one image per forward
    and for batches of equally shaped images; the golden detections of the whole demo set come out of it"""
    import os
    import numpy as np
    import torch
    import torch.distributed as dist
    from kgdet_amd import runner
    cfg, model = demo_cases.demo_detector()
    model = model.cuda()
    data = demo_cases.demo_dataset(test_mode=True)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29577')
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        to_dev = lambda t: t.cuda(non_blocking=True)
        one = runner.multi_gpu_test(model, data, rescale=True, to_device=to_dev)
        batched = runner.multi_gpu_test(model, data, rescale=True, to_device=to_dev, imgs_per_gpu=4)
    finally:
        dist.destroy_process_group()
    want = demo_checks.run_detector(model, data, 'cuda', list(range(len(data))))
    assert len(one) == len(batched) == len(data) == 32
    G = np.load(demo_checks.GOLDEN)
    demo_checks.compare_detections({i: r for i, r in enumerate(one)}, data, G)
    n = 0
    for i in range(len(data)):
        a, b, w = one[i], batched[i], want[i]
        assert len(a) == len(b) == len(w)
        if len(w) == 3:
            for x, y, z in zip(a[0], b[0], w[0]):
                assert np.array_equal(x, z) and x.shape == y.shape and np.allclose(x, y, rtol=1e-5, atol=1e-4)
                n += len(z)
    assert n == 401
