"""BASELINE config 1 on the CPU (plumbing, no GPU): demo annotations -> dataset pipeline -> this repo's detector with
the test-side CPU ops -> writers -> evaluator, against the REFERENCE detector's detections / AP on the same rendered
images (tests/golden/demo_dets_golden.npz).  A subset of the 32 images keeps the CPU suite short; the GPU suite runs
all of them through the HIP detector (tests/test_gpu_demo_set.py)."""
import numpy as np

from tests import cpu_ops, demo_checks
from tests.golden import demo_cases


def test_rendered_demo_images_are_reproducible():
    data = demo_cases.demo_dataset(test_mode=True)
    a, b = data.load_image(3), data.load_image(3)
    assert a.dtype == np.uint8 and a.shape == (data.img_infos[3]['height'], data.img_infos[3]['width'], 3)
    assert np.array_equal(a, b) and not np.array_equal(a, data.load_image(4)[:a.shape[0], :a.shape[1]])


def test_demo_subset_cpu_path_equals_reference_detector(tmp_path):
    cfg, model = demo_cases.demo_detector()
    with cpu_ops.patched():
        out = demo_checks.check_demo_set(model, 'cpu', [0, 9, 13, 19], tmp_path, full_set=False)
    print(out)
    assert out['detections'] == 8 + 0 + 28 + 48
