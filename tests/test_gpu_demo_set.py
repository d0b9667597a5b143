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
