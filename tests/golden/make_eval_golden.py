"""Generate tests/golden/eval_golden.npz with the REFERENCE evaluator (run in the build container only -- needs
/root/reference):

    python tests/golden/make_eval_golden.py

Ground truth = the reference's demo annotations (data/demo_dataset/demo_dataset-32.json, copied next to this file as a
data fixture).  Detections = seeded perturbations of the ground truth (jittered duplicates, wrong categories, pure
false positives), rounded to 4 digits as kpt2json writes them.  Expected outputs = what the unmodified
deepfashion2_api COCOeval (imported in place by oracle/build_ref.py) returns: the summary stats and the full
precision / recall arrays, for 'bbox' and 'keypoints', plus the ground-truth-as-detections known-answer case."""
import io
import json
import os
import shutil
import sys
import tempfile
from contextlib import redirect_stdout

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402

GT_SRC = '/root/reference/data/demo_dataset/demo_dataset-32.json'
GT_FIX = os.path.join(HERE, 'demo_dataset-32.json')


def synth_detections(gt, seed):
    rng = np.random.default_rng(seed)
    cat_ids = [c['id'] for c in gt['categories']]
    img_sizes = {im['id']: (im['width'], im['height']) for im in gt['images']}
    boxes, kpts, cats, imgs, scores = [], [], [], [], []
    for ann in gt['annotations']:
        x, y, w, h = ann['bbox']
        g = np.asarray(ann['keypoints'], dtype=np.float64).reshape(-1, 3)
        for _ in range(int(rng.integers(0, 4))):
            noise = float(rng.choice([0.01, 0.05, 0.15, 0.4]))
            b = np.array([x, y, w, h]) + rng.normal(0, noise, 4) * np.array([w, h, w, h])
            b[2:] = np.maximum(b[2:], 1.0)
            k = g.copy()
            vis = k[:, 2] > 0
            k[vis, :2] += rng.normal(0, noise * 0.3 * np.sqrt(ann['area']), (int(vis.sum()), 2))
            k[vis, 2] = 1.0
            cat = ann['category_id'] if rng.random() > 0.15 else int(rng.choice(cat_ids))
            boxes.append(b); kpts.append(k.reshape(-1)); cats.append(cat); imgs.append(ann['image_id'])
            scores.append(float(rng.random()))
    for _ in range(25):   # pure false positives
        img_id = int(rng.choice(list(img_sizes)))
        W, H = img_sizes[img_id]
        b = np.array([rng.uniform(0, W / 2), rng.uniform(0, H / 2), rng.uniform(10, W / 2), rng.uniform(10, H / 2)])
        k = np.zeros((294, 3))
        sel = rng.choice(294, 20, replace=False)
        k[sel, 0], k[sel, 1], k[sel, 2] = rng.uniform(0, W, 20), rng.uniform(0, H, 20), 1.0
        boxes.append(b); kpts.append(k.reshape(-1)); cats.append(int(rng.choice(cat_ids))); imgs.append(img_id)
        scores.append(float(rng.random() * 0.6))
    return (np.round(np.array(boxes), 4), np.round(np.array(kpts), 4), np.array(cats), np.array(imgs),
            np.round(np.array(scores), 4))


def as_results(boxes, kpts, cats, imgs, scores):
    b = [dict(image_id=int(i), bbox=[float(v) for v in bb], score=float(s), category_id=int(c))
         for bb, c, i, s in zip(boxes, cats, imgs, scores)]
    k = [dict(image_id=int(i), keypoints=[float(v) for v in kk], score=float(s), category_id=int(c))
         for kk, c, i, s in zip(kpts, cats, imgs, scores)]
    return b, k


def run_reference(COCO, COCOeval, gt_file, results, iou_type):
    with redirect_stdout(io.StringIO()):
        gt = COCO(gt_file)
        with tempfile.NamedTemporaryFile('w', suffix='.json', delete=False) as f:   # loadRes takes a file name
            json.dump(results, f)
        dt = gt.loadRes(f.name)
        os.unlink(f.name)
        ev = COCOeval(gt, dt, iou_type)
        ev.params.imgIds = gt.getImgIds()
        ev.evaluate(); ev.accumulate(); ev.summarize()
    return ev.stats, ev.eval['precision'], ev.eval['recall']


def main():
    ref = build_ref.load_reference_evaluator()
    assert ref is not None, 'reference checkout not present'
    COCO, COCOeval = ref
    shutil.copyfile(GT_SRC, GT_FIX)
    os.chmod(GT_FIX, 0o644)
    gt = json.load(open(GT_FIX))
    out = {}
    for case, seed in (('a', 0), ('b', 7)):
        boxes, kpts, cats, imgs, scores = synth_detections(gt, seed)
        out.update({'%s_boxes' % case: boxes, '%s_kpts' % case: kpts, '%s_cats' % case: cats, '%s_imgs' % case: imgs,
                    '%s_scores' % case: scores})
        b, k = as_results(boxes, kpts, cats, imgs, scores)
        for typ, res in (('bbox', b), ('keypoints', k)):
            stats, prec, rec = run_reference(COCO, COCOeval, GT_FIX, res, typ)
            out['%s_%s_stats' % (case, typ)] = stats
            out['%s_%s_precision' % (case, typ)] = prec
            out['%s_%s_recall' % (case, typ)] = rec
            print(case, typ, np.round(stats, 4))
    # known answer: ground truth fed back as detections
    b = [dict(image_id=a['image_id'], bbox=a['bbox'], score=1.0, category_id=a['category_id']) for a in gt['annotations']]
    k = [dict(image_id=a['image_id'], keypoints=a['keypoints'], score=1.0, category_id=a['category_id'])
         for a in gt['annotations']]
    for typ, res in (('bbox', b), ('keypoints', k)):
        stats, _, _ = run_reference(COCO, COCOeval, GT_FIX, res, typ)
        out['gt_%s_stats' % typ] = stats
        print('gt', typ, np.round(stats, 4))
    np.savez_compressed(os.path.join(HERE, 'eval_golden.npz'), **out)


if __name__ == '__main__':
    main()
