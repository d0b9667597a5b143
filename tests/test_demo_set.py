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


def _perturbed(G, data, indices, rel, drop_every=0, seed=0):
    """the reference's detections as a second "model": coordinates moved by `rel` of the image size (boxes AND landmarks),
    optionally every `drop_every`-th detection removed"""
    rng = np.random.default_rng(seed)
    H = {}
    for k in G.files:
        H[k] = G[k].copy()
    for idx in indices:
        gb, gk = H['img%d:bboxes' % idx], H['img%d:kpts' % idx]
        if not len(gb):
            continue
        scale = float(max(data.img_infos[idx]['width'], data.img_infos[idx]['height']))
        gb[:, :4] += rng.uniform(-1, 1, size=gb[:, :4].shape).astype(np.float32) * rel * scale
        xy = gk.reshape(len(gk), -1, 3)
        live = xy[:, :, 2] != 0
        xy[:, :, :2] += (rng.uniform(-1, 1, size=xy[:, :, :2].shape) * rel * scale * live[:, :, None]).astype(np.float32)
        if drop_every:
            keep = np.arange(len(gb)) % drop_every != 0
            H['img%d:bboxes' % idx], H['img%d:labels' % idx], H['img%d:kpts' % idx] = gb[keep], H['img%d:labels' % idx][keep], gk[keep]
    return H


def test_ap_criterion_is_not_vacuous(tmp_path):
    """north star: "bbox / keypoint AP on the demo set within +-0.1 of the reference".  With random-init weights both APs
    against the demo ANNOTATIONS are 0.000 (no checkpoint travels), so that comparison alone could not fail.  This test
    shows what the criterion does on NON-ZERO APs, with the reference's own 401 demo-set detections
    (demo_dets_golden.npz) as ground truth and perturbed copies of them as "models":
      * a model at the MEASURED parity level (coordinates within 2e-5 relative -- the HIP detector is at 1.3e-5 --, same
        selection) keeps both APs within +-0.1 points of the attainable value;
      * a model at the EDGE of the north star's coordinate tolerance (1e-3 relative = 1.3 px) already moves the box AP by
        ~11 points (AP averages IoU thresholds up to 0.95): of the two parity statements the AP one is the stricter, and a
        pass of the coordinate check at 1e-3 would not imply it;
      * a model 2 % off in its coordinates, or one that loses every fifth detection, moves both APs by far more than 0.1
        points -- the criterion rejects them."""
    G = np.load(demo_checks.GOLDEN)
    data = demo_cases.demo_dataset(test_mode=True)
    indices = list(range(len(data)))
    gt = demo_checks.golden_as_ground_truth(G, data, indices)

    def ap(H, tag):
        st = demo_checks.evaluate(demo_checks.golden_as_results(H, data, indices), data, indices, gt, tmp_path, tag)
        return {t: float(st[t][0]) * 100 for t in ('bbox', 'keypoints')}
    best = ap(G, 'exact')
    assert best['bbox'] > 50 and best['keypoints'] > 50, best
    near = ap(_perturbed(G, data, indices, 2e-5), 'near')
    edge = ap(_perturbed(G, data, indices, 1e-3), 'edge')
    far = ap(_perturbed(G, data, indices, 2e-2), 'far')
    lossy = ap(_perturbed(G, data, indices, 0.0, drop_every=5), 'lossy')
    print('AP (bbox, keypoints) exact / 2e-5 / 1e-3 / 2 %% off / every fifth lost:', best, near, edge, far, lossy)
    assert best['bbox'] - edge['bbox'] > 0.1, (edge, best)
    for t in ('bbox', 'keypoints'):
        assert abs(near[t] - best[t]) <= 0.1, (t, near, best)
        assert best[t] - lossy[t] > 0.1, (t, lossy, best)
    assert best['bbox'] - far['bbox'] > 0.1 and best['keypoints'] - far['keypoints'] > 0.1, (far, best)
