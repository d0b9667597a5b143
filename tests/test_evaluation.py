"""kgdet_amd.evaluation against the reference's DeepFashion2 evaluator.

Golden vectors: tests/golden/eval_golden.npz, written by tests/golden/make_eval_golden.py from the unmodified
deepfashion2_api COCOeval; ground truth = the demo annotation file (data fixture next to it)."""
import copy
import json
import os

import numpy as np
import pytest

from kgdet_amd import evaluation as ev

HERE = os.path.dirname(os.path.abspath(__file__))
GT = os.path.join(HERE, 'golden', 'demo_dataset-32.json')
GOLD = np.load(os.path.join(HERE, 'golden', 'eval_golden.npz'))


def _results(case):
    boxes, kpts = GOLD[case + '_boxes'], GOLD[case + '_kpts']
    cats, imgs, scores = GOLD[case + '_cats'], GOLD[case + '_imgs'], GOLD[case + '_scores']
    b = [dict(image_id=int(i), bbox=[float(v) for v in bb], score=float(s), category_id=int(c))
         for bb, c, i, s in zip(boxes, cats, imgs, scores)]
    k = [dict(image_id=int(i), keypoints=[float(v) for v in kk], score=float(s), category_id=int(c))
         for kk, c, i, s in zip(kpts, cats, imgs, scores)]
    return dict(bbox=b, keypoints=k)


def _run(results, typ):
    gt = ev.CocoIndex(GT)
    e = ev.CocoEvaluator(gt, gt.load_results(copy.deepcopy(results)), typ)
    e.params.img_ids = gt.get_img_ids()
    e.evaluate().accumulate()
    return e.summarize(verbose=False), e.eval['precision'], e.eval['recall']


@pytest.mark.parametrize('case', ['a', 'b'])
@pytest.mark.parametrize('typ', ['bbox', 'keypoints'])
def test_evaluator_matches_reference_golden(case, typ):
    stats, prec, rec = _run(_results(case)[typ], typ)
    np.testing.assert_allclose(stats, GOLD['%s_%s_stats' % (case, typ)], rtol=0, atol=1e-12)
    np.testing.assert_allclose(prec, GOLD['%s_%s_precision' % (case, typ)], rtol=0, atol=1e-12)
    np.testing.assert_allclose(rec, GOLD['%s_%s_recall' % (case, typ)], rtol=0, atol=1e-12)
    assert (stats[[0, 1, 2]] > 0.1).all() and (stats[[0, 1, 2]] < 0.9).all()   # a non-trivial case


def test_ground_truth_as_detections_known_answer():
    """the 55 demo instances fed back with score 1 give AP = 1 (bbox AR@1 = 0.950)."""
    gt = json.load(open(GT))
    b = [dict(image_id=a['image_id'], bbox=a['bbox'], score=1.0, category_id=a['category_id']) for a in gt['annotations']]
    k = [dict(image_id=a['image_id'], keypoints=a['keypoints'], score=1.0, category_id=a['category_id'])
         for a in gt['annotations']]
    sb, _, _ = _run(b, 'bbox')
    sk, _, _ = _run(k, 'keypoints')
    np.testing.assert_allclose(sb, GOLD['gt_bbox_stats'], atol=1e-12)
    np.testing.assert_allclose(sk, GOLD['gt_keypoints_stats'], atol=1e-12)
    assert sb[0] == 1.0 and abs(sb[6] - 0.95) < 1e-12 and sk[0] == 1.0


def test_live_reference_evaluator_agrees_when_present():
    from oracle import build_ref
    ref = build_ref.load_reference_evaluator()
    if ref is None:
        pytest.skip('reference checkout not present')
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_eval_golden', os.path.join(HERE, 'golden', 'make_eval_golden.py'))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    gt = json.load(open(GT))
    b, k = gen.as_results(*gen.synth_detections(gt, seed=123))
    for typ, res in (('bbox', b), ('keypoints', k)):
        want, wp, wr = gen.run_reference(ref[0], ref[1], GT, res, typ)
        got, gp, gr = _run(res, typ)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
        np.testing.assert_allclose(gp, wp, rtol=0, atol=1e-12)
        np.testing.assert_allclose(gr, wr, rtol=0, atol=1e-12)


class _FakeDataset(object):
    img_ids = [11, 12]
    cat_ids = [1, 2, 3]

    def __len__(self):
        return 2


def test_kpt2json_rounding_and_layout(tmp_path):
    """coco_utils.py:121-154: +1 widths, 4-digit rounding, landmark score = box score, one entry per detection."""
    det = np.array([[10.12345, 20.5, 30.98765, 60.25, 0.87654321]], np.float32)
    kp = (np.arange(882, dtype=np.float32) * 0.123456).reshape(1, 882)
    empty = np.zeros((0, 5), np.float32)
    results = [([det, empty, empty], [det[:, 4], empty[:, 4], empty[:, 4]], [kp, np.zeros((0, 882)), np.zeros((0, 882))]),
               ([empty, empty, det], [empty[:, 4], empty[:, 4], det[:, 4]], [np.zeros((0, 882)), np.zeros((0, 882)), kp])]
    boxes, kpts = ev.kpt2json(_FakeDataset(), results)
    assert [b['image_id'] for b in boxes] == [11, 12] and [b['category_id'] for b in boxes] == [1, 3]
    x1, y1, x2, y2 = [float(v) for v in det[0, :4]]
    assert boxes[0]['bbox'] == [round(x1, 4), round(y1, 4), round(x2 - x1 + 1, 4), round(y2 - y1 + 1, 4)]
    assert boxes[0]['score'] == round(float(det[0, 4]), 4) == kpts[0]['score']
    assert kpts[1]['keypoints'] == np.round(kp[0].astype(np.float64), 4).tolist() and len(kpts[1]['keypoints']) == 882
    files = ev.results2json(_FakeDataset(), results, str(tmp_path / 'res'))
    assert sorted(files) == ['bbox', 'keypoints', 'proposal']
    assert json.load(open(files['keypoints']))[0]['keypoints'] == kpts[0]['keypoints']
    with pytest.raises(TypeError):
        ev.results2json(_FakeDataset(), [np.zeros((1, 5))], str(tmp_path / 'x'))


def test_detections2result_splits_by_class():
    b = np.array([[0, 0, 1, 1, .9], [0, 0, 2, 2, .8], [0, 0, 3, 3, .7]], np.float32)
    det, score, kpt = ev.detections2result(b, np.array([2, 0, 2]), np.arange(3 * 882).reshape(3, 882), num_classes=14)
    assert len(det) == 13 and det[2].shape == (2, 5) and det[0].shape == (1, 5) and det[1].shape == (0, 5)
    assert kpt[2].shape == (2, 882) and float(score[0][0]) == np.float32(.8)


def test_box_iou_and_oks_edge_cases():
    iou = ev.box_iou_xywh([[0, 0, 10, 10], [20, 20, 5, 5]], [[5, 5, 10, 10], [0, 0, 10, 10]], [0, 1])
    assert abs(iou[0, 0] - 25.0 / 175.0) < 1e-15 and iou[0, 1] == 1.0 and iou[1, 0] == 0.0
    sig = ev.landmark_meta()['oks_sigmas']
    assert sig.shape == (294,) and abs(sig[0] - 0.012) < 1e-12
    kp = np.zeros(882)
    gt = dict(keypoints=kp.tolist(), bbox=[10, 10, 20, 20], area=400.0)     # nothing labelled -> box distance rule
    d_in = np.zeros(882); d_in[0::3] = 15; d_in[1::3] = 15
    assert ev.oks([d_in], [gt], sig)[0, 0] == 1.0
