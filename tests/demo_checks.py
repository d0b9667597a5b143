"""BASELINE config 1 / north-star "AP on the demo set": dataset pipeline -> detector -> result writers -> evaluator,
compared with tests/golden/demo_dets_golden.npz (the REFERENCE's detector + writers + COCOeval on its CPU path,
tests/golden/make_demo_golden.py).  Device-agnostic (CPU: test-side ops, GPU: the HIP ops)."""
import os

import numpy as np
import torch

from kgdet_amd import evaluation
from tests.golden import demo_cases

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'demo_dets_golden.npz')


def run_detector(model, data, device, indices):
    """tools/test.py:19-35 single_gpu_test: one image at a time, rescale=True"""
    model.eval()
    results = {}
    with torch.no_grad():
        for idx in indices:
            d = data[idx]
            results[idx] = model(return_loss=False, rescale=True, img=[d['img'][0][None].to(device)],
                                 img_meta=[[d['img_meta'][0]]])
    return results


def flatten(result, data):
    if len(result) != 3:
        return np.zeros((0, 5), np.float32), np.zeros(0, np.int64), np.zeros((0, 117), np.float32)
    boxes = np.concatenate(result[0]).astype(np.float32)
    labels = np.concatenate([np.full(len(b), c, np.int64) for c, b in enumerate(result[0])])
    kpts = np.concatenate(result[2]).astype(np.float32)
    sl = demo_cases.category_slices(data)
    k = np.stack([np.pad(kk[3 * sl[c + 1][0]:3 * sl[c + 1][1]], (0, 117))[:117] for kk, c in zip(kpts, labels)]) \
        if len(labels) else np.zeros((0, 117), np.float32)
    return boxes, labels, k


def compare_detections(results, data, G, tol=1e-3):
    """north star: NMS selection identical (same detections, same classes, same order), coordinates 1e-3 relative"""
    worst = 0.0
    for idx, r in results.items():
        boxes, labels, kpts = flatten(r, data)
        gb, gl, gk = G['img%d:bboxes' % idx], G['img%d:labels' % idx], G['img%d:kpts' % idx]
        assert boxes.shape == gb.shape, (idx, boxes.shape, gb.shape)
        assert np.array_equal(labels, gl), idx
        if len(gl):
            scale = max(float(np.abs(gb[:, :4]).max()), 1.0)
            worst = max(worst, float(np.abs(boxes[:, :4] - gb[:, :4]).max()) / scale,
                        float(np.abs(kpts - gk).max()) / max(float(np.abs(gk).max()), 1.0),
                        float(np.abs(boxes[:, 4] - gb[:, 4]).max()) / max(float(gb[:, 4].max()), 1e-6))
    assert worst < tol, worst
    return worst


def golden_as_results(G, data, indices):
    """the golden detections in the per-class tuple form of bbox2result_kp (landmarks back in their 882 slots)"""
    sl = demo_cases.category_slices(data)
    out = {}
    for idx in indices:
        gb, gl, gk = G['img%d:bboxes' % idx], G['img%d:labels' % idx], G['img%d:kpts' % idx]
        if not len(gl):
            out[idx] = ([np.zeros((0, 5), np.float32) for _ in range(13)], )
            continue
        full = np.zeros((len(gl), 882), np.float32)
        for i, c in enumerate(gl):
            lo, hi = sl[c + 1]
            full[i, 3 * lo:3 * hi] = gk[i, :3 * (hi - lo)]
        out[idx] = evaluation.detections2result(gb, gl, full, 14)
    return out


def golden_as_ground_truth(G, data, indices):
    """COCO-style annotation dict whose ground truth IS the golden detection set (so AP measures detection parity)"""
    sl = demo_cases.category_slices(data)
    images, anns = [], []
    for idx in indices:
        info = data.img_infos[idx]
        images.append(dict(id=info['id'], width=info['width'], height=info['height'], file_name=info['file_name']))
        gb, gl, gk = G['img%d:bboxes' % idx], G['img%d:labels' % idx], G['img%d:kpts' % idx]
        for i, c in enumerate(gl):
            lo, hi = sl[c + 1]
            kp = np.zeros(882, np.float64)
            seg = gk[i, :3 * (hi - lo)].astype(np.float64).copy()
            seg[2::3] = 2
            kp[3 * lo:3 * hi] = seg
            x1, y1, x2, y2 = [float(v) for v in gb[i, :4]]
            anns.append(dict(id=len(anns) + 1, image_id=info['id'], category_id=data.cat_ids[c], iscrowd=0,
                             bbox=[x1, y1, x2 - x1 + 1, y2 - y1 + 1], area=(x2 - x1 + 1) * (y2 - y1 + 1),
                             keypoints=kp.tolist(), num_keypoints=int(hi - lo)))
    return dict(images=images, annotations=anns, categories=[dict(id=c, name=str(c)) for c in data.cat_ids])


class _Subset(object):
    def __init__(self, data, indices):
        self.img_ids = [data.img_ids[i] for i in indices]
        self.cat_ids = data.cat_ids
        self._n = len(indices)

    def __len__(self):
        return self._n


def evaluate(results, data, indices, gt, tmpdir, tag):
    """results2json + coco_eval (coco_utils.py:121-216 flow) -> {'bbox': stats, 'keypoints': stats}"""
    ordered = [results[i] for i in indices]
    files = evaluation.results2json(_Subset(data, indices), ordered, os.path.join(str(tmpdir), tag))
    if not os.path.exists(files.get('keypoints', '')) or all(len(r) != 3 for r in ordered):
        return None
    coco = gt if isinstance(gt, evaluation.CocoIndex) else evaluation.CocoIndex(gt)
    return evaluation.coco_eval(files, ['bbox', 'keypoints'], coco, verbose=False)


def check_demo_set(model, device, indices, tmpdir, full_set):
    G = np.load(GOLDEN)
    data = demo_cases.demo_dataset(test_mode=True)
    results = run_detector(model, data, device, indices)
    worst = compare_detections(results, data, G)
    out = dict(worst_rel=worst, detections=sum(len(flatten(r, data)[1]) for r in results.values()))
    # (1) AP against the demo annotations, as tools/test.py reports it, next to the reference's number
    if full_set:
        stats = evaluate(results, data, indices, data.coco, tmpdir, 'demo')
        for typ in ('bbox', 'keypoints'):
            ref = G['stats:' + typ]
            assert np.all(np.abs(stats[typ] * 100 - ref * 100) <= 0.1), (typ, stats[typ], ref)     # AP within +-0.1
        out['ap_vs_demo_gt'] = {t: float(stats[t][0]) for t in stats}
    # (2) AP with the REFERENCE's detections as ground truth: 1.000-level agreement is what "same detections" means;
    #     the golden evaluated against itself gives the attainable value (maxDets = 20 caps the keypoint recall)
    gt = golden_as_ground_truth(G, data, indices)
    mine = evaluate(results, data, indices, gt, tmpdir, 'mine')
    best = evaluate(golden_as_results(G, data, indices), data, indices, gt, tmpdir, 'golden')
    for typ in ('bbox', 'keypoints'):
        assert np.all(np.abs(mine[typ] - best[typ]) <= 1e-3), (typ, mine[typ], best[typ])
        assert best[typ][0] > 0.5
    out['ap_vs_reference_detections'] = {t: (float(mine[t][0]), float(best[t][0])) for t in mine}
    return out
