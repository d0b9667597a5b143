"""Result writers and the DeepFashion2 bbox / landmark (OKS) evaluator  (SURVEY 8f row 2).

Host-side numpy code, no compiled extension: the reference does this part on the CPU too.

* ``CocoIndex`` -- the slice of the COCO API the path uses (``deepfashion2_api/PythonAPI/pycocotools/coco.py``:
  ``createIndex`` :84-116, ``getAnnIds`` :130-158, ``getCatIds`` :160-181, ``getImgIds`` :183-203, ``loadRes``
  :297-356: a bbox result gets ``area = w * h``; a landmark result gets ``bbox`` / ``area`` from the extent of
  ALL its coordinates).
* ``det2json`` / ``kpt2json`` / ``results2json`` -- ``mmdet/core/evaluation/coco_utils.py:104-216``: xyxy -> xywh
  with the +1 width convention, 4-digit rounding in ``kpt2json``, landmark score = the box score.
* ``CocoEvaluator`` -- ``pycocotools/cocoeval.py``: per (image, category) similarity matrix (box IoU as
  ``maskUtils.iou`` computes it for [x, y, w, h] boxes; OKS with the 294 DeepFashion2 sigmas, :193-271), greedy
  score-ordered matching per IoU threshold with crowd / ignore rules (:273-351), 101-point interpolated precision
  (:353-460), and the 12 (bbox) / 10 (landmark, ``maxDets = 20``) summary numbers (:462-538).
  Pinned against the compiled reference evaluator by ``tests/golden/eval_golden.json``.
"""
import json
import os
from collections import defaultdict

import numpy as np

_META = None


def landmark_meta():
    """DeepFashion2's 294-landmark scheme (kgdet_amd/data/deepfashion2_landmarks.json): class names, per-category
    landmark ranges, per-category left/right swap pairs, landmark groups, OKS sigmas -- all 0-based global indices."""
    global _META
    if _META is None:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'deepfashion2_landmarks.json')) as f:
            m = json.load(f)
        m['oks_sigmas'] = np.asarray(m.pop('oks_sigmas_e4'), dtype=np.float64) * 1e-4
        _META = m
    return _META


class CocoIndex(object):
    """COCO-format annotations indexed by image / category."""

    def __init__(self, annotation=None):
        self.dataset = {}
        self.anns, self.imgs, self.cats = {}, {}, {}
        self.img_to_anns, self.cat_to_imgs = defaultdict(list), defaultdict(list)
        if annotation is not None:
            if isinstance(annotation, (str, bytes, os.PathLike)):
                with open(annotation) as f:
                    annotation = json.load(f)
            if not isinstance(annotation, dict):
                raise TypeError('annotation file format {} not supported'.format(type(annotation)))
            self.dataset = annotation
            self.create_index()

    def create_index(self):
        self.anns, self.imgs, self.cats = {}, {}, {}
        self.img_to_anns, self.cat_to_imgs = defaultdict(list), defaultdict(list)
        for ann in self.dataset.get('annotations', ()):
            self.img_to_anns[ann['image_id']].append(ann)
            self.anns[ann['id']] = ann
        for img in self.dataset.get('images', ()):
            self.imgs[img['id']] = img
        for cat in self.dataset.get('categories', ()):
            self.cats[cat['id']] = cat
        if 'categories' in self.dataset:
            for ann in self.dataset.get('annotations', ()):
                self.cat_to_imgs[ann['category_id']].append(ann['image_id'])

    # --- queries (argument conventions of the COCO API: empty filter = everything) ---
    def get_ann_ids(self, img_ids=(), cat_ids=()):
        img_ids = list(img_ids) if isinstance(img_ids, (list, tuple, set, np.ndarray)) else [img_ids]
        cat_ids = list(cat_ids) if isinstance(cat_ids, (list, tuple, set, np.ndarray)) else [cat_ids]
        if img_ids:
            anns = [a for i in img_ids if i in self.img_to_anns for a in self.img_to_anns[i]]
        else:
            anns = self.dataset.get('annotations', [])
        if cat_ids:
            keep = set(cat_ids)
            anns = [a for a in anns if a['category_id'] in keep]
        return [a['id'] for a in anns]

    def get_cat_ids(self):
        return [c['id'] for c in self.dataset.get('categories', ())]

    def get_img_ids(self):
        return list(self.imgs.keys())

    def load_anns(self, ids):
        return [self.anns[i] for i in ids]

    def load_imgs(self, ids):
        return [self.imgs[i] for i in ids]

    def load_results(self, results):
        """Detections (list of dicts or a json file of them) -> a CocoIndex over the same images."""
        if isinstance(results, (str, bytes, os.PathLike)):
            with open(results) as f:
                results = json.load(f)
        if not isinstance(results, list):
            raise TypeError('results must be a list of objects')
        res = CocoIndex()
        res.dataset['images'] = list(self.dataset['images'])
        known = set(self.get_img_ids())
        if any(r['image_id'] not in known for r in results):
            raise ValueError('Results do not correspond to current coco set')
        if results:
            res.dataset['categories'] = json.loads(json.dumps(self.dataset['categories']))
            first = results[0]
            if 'bbox' in first and first['bbox'] != []:
                for k, r in enumerate(results):
                    r['area'] = r['bbox'][2] * r['bbox'][3]
                    r['id'] = k + 1
                    r['iscrowd'] = 0
            elif 'keypoints' in first:
                for k, r in enumerate(results):
                    xs, ys = r['keypoints'][0::3], r['keypoints'][1::3]
                    x0, x1, y0, y1 = np.min(xs), np.max(xs), np.min(ys), np.max(ys)
                    r['area'] = (x1 - x0) * (y1 - y0)
                    r['id'] = k + 1
                    r['bbox'] = [x0, y0, x1 - x0, y1 - y0]
            else:
                raise ValueError('only bbox and keypoints results are supported')
        res.dataset['annotations'] = results
        res.create_index()
        return res


# ------------------------------------------------------------------------------------------------
# result writers
# ------------------------------------------------------------------------------------------------
def xyxy2xywh(bbox):
    b = np.asarray(bbox).tolist()
    return [b[0], b[1], b[2] - b[0] + 1, b[3] - b[1] + 1]


def det2json(dataset, results):
    out = []
    for idx in range(len(dataset)):
        for label, boxes in enumerate(results[idx]):
            for row in boxes:
                out.append(dict(image_id=dataset.img_ids[idx], bbox=xyxy2xywh(row), score=float(row[4]),
                                category_id=dataset.cat_ids[label]))
    return out


def kpt2json(dataset, results, num_digits=4):
    """``results[idx] = (per-class boxes [n, 5], per-class scores, per-class landmarks [n, 3 * 294])``."""
    box_out, kpt_out = [], []
    for idx in range(len(dataset)):
        if len(results[idx]) != 3:
            continue
        det, _, kpt = results[idx]
        img_id = dataset.img_ids[idx]
        for label in range(len(det)):
            boxes, cat = det[label], dataset.cat_ids[label]
            for row in boxes:
                box_out.append(dict(image_id=img_id, bbox=[round(v, num_digits) for v in xyxy2xywh(row)],
                                    score=round(float(row[4]), num_digits), category_id=cat))
            for i, pts in enumerate(kpt[label]):
                kpt_out.append(dict(image_id=img_id,
                                    keypoints=np.round(np.asarray(pts).astype(np.float64), num_digits).tolist(),
                                    score=round(float(boxes[i][4]), num_digits), category_id=cat))
    return box_out, kpt_out


def results2json(dataset, results, out_file):
    files = {}
    if isinstance(results[0], list):
        files['bbox'] = files['proposal'] = '{}.bbox.json'.format(out_file)
        with open(files['bbox'], 'w') as f:
            json.dump(det2json(dataset, results), f)
    elif isinstance(results[0], tuple):
        boxes, kpts = kpt2json(dataset, results)
        files['bbox'] = files['proposal'] = '{}.bbox.json'.format(out_file)
        files['keypoints'] = '{}.keypoints.json'.format(out_file)
        with open(files['bbox'], 'w') as f:
            json.dump(boxes, f)
        with open(files['keypoints'], 'w') as f:
            json.dump(kpts, f)
    else:
        raise TypeError('invalid type of results')
    return files


def detections2result(det_bboxes, det_labels, det_kpts, num_classes):
    """One image's detections (label 0-based) -> the per-class tuple ``kpt2json`` takes
    (``bbox2result_kp``-style splitting, mmdet/core/bbox/transforms.py:138-156 applied to all three arrays)."""
    b = np.asarray(det_bboxes, dtype=np.float32).reshape(-1, 5)
    lab = np.asarray(det_labels).reshape(-1)
    k = np.asarray(det_kpts, dtype=np.float32).reshape(b.shape[0], -1)
    return ([b[lab == c] for c in range(num_classes - 1)], [b[lab == c, 4] for c in range(num_classes - 1)],
            [k[lab == c] for c in range(num_classes - 1)])


# ------------------------------------------------------------------------------------------------
# evaluator
# ------------------------------------------------------------------------------------------------
def _linspace_thresholds(lo, hi, step):
    return np.linspace(lo, hi, int(np.round((hi - lo) / step)) + 1, endpoint=True)


class EvalParams(object):
    def __init__(self, iou_type):
        if iou_type not in ('bbox', 'keypoints'):
            raise ValueError('iou_type {!r} not supported'.format(iou_type))
        self.iou_type = iou_type
        self.img_ids, self.cat_ids = [], []
        self.iou_thrs = _linspace_thresholds(0.5, 0.95, 0.05)
        self.rec_thrs = _linspace_thresholds(0.0, 1.0, 0.01)
        self.use_cats = 1
        if iou_type == 'bbox':
            self.max_dets = [1, 10, 100]
            self.area_rng = [[0, 1e10], [0, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e10]]
            self.area_lbl = ['all', 'small', 'medium', 'large']
        else:
            self.max_dets = [20]
            self.area_rng = [[0, 1e10], [32 ** 2, 96 ** 2], [96 ** 2, 1e10]]
            self.area_lbl = ['all', 'medium', 'large']


def box_iou_xywh(dts, gts, crowd):
    """[D, G] IoU of [x, y, w, h] boxes; against a crowd ground truth the union is the detection's own area."""
    d = np.asarray(dts, dtype=np.float64).reshape(-1, 4)
    g = np.asarray(gts, dtype=np.float64).reshape(-1, 4)
    iw = np.minimum(d[:, None, 0] + d[:, None, 2], g[None, :, 0] + g[None, :, 2]) - np.maximum(d[:, None, 0], g[None, :, 0])
    ih = np.minimum(d[:, None, 1] + d[:, None, 3], g[None, :, 1] + g[None, :, 3]) - np.maximum(d[:, None, 1], g[None, :, 1])
    inter = np.where((iw > 0) & (ih > 0), iw * ih, 0.0)
    da, ga = (d[:, 2] * d[:, 3])[:, None], (g[:, 2] * g[:, 3])[None, :]
    union = np.where(np.asarray(crowd, dtype=bool)[None, :], da, da + ga - inter)
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.where(inter > 0, inter / union, 0.0)


def oks(dts, gts, sigmas):
    """[D, G] object-keypoint similarity.  ``dts``: landmark triplets [D, 3k]; ``gts``: list of annotations."""
    var = (np.asarray(sigmas, dtype=np.float64) * 2) ** 2
    d = np.asarray(dts, dtype=np.float64).reshape(len(dts), -1)
    xd, yd = d[:, 0::3], d[:, 1::3]
    out = np.zeros((d.shape[0], len(gts)))
    for j, gt in enumerate(gts):
        g = np.asarray(gt['keypoints'], dtype=np.float64)
        xg, yg, vg = g[0::3], g[1::3], g[2::3]
        vis = vg > 0
        if vis.any():
            dx, dy = xd - xg, yd - yg
        else:   # no labelled landmark: distance to the doubled ground-truth box
            bx, by, bw, bh = gt['bbox']
            x0, x1, y0, y1 = bx - bw, bx + bw * 2, by - bh, by + bh * 2
            dx = np.maximum(0, x0 - xd) + np.maximum(0, xd - x1)
            dy = np.maximum(0, y0 - yd) + np.maximum(0, yd - y1)
        e = (dx ** 2 + dy ** 2) / var / (gt['area'] + np.spacing(1)) / 2
        if vis.any():
            e = e[:, vis]
        out[:, j] = np.exp(-e).sum(axis=1) / e.shape[1]
    return out


class CocoEvaluator(object):
    def __init__(self, gt, dt, iou_type):
        self.gt, self.dt = gt, dt
        self.params = EvalParams(iou_type)
        self.params.img_ids = sorted(gt.get_img_ids())
        self.params.cat_ids = sorted(gt.get_cat_ids())
        self.eval_imgs, self.eval, self.stats = [], {}, None

    def _prepare(self):
        p = self.params
        sel = dict(img_ids=p.img_ids, cat_ids=p.cat_ids) if p.use_cats else dict(img_ids=p.img_ids)
        gts = self.gt.load_anns(self.gt.get_ann_ids(**sel))
        dts = self.dt.load_anns(self.dt.get_ann_ids(**sel))
        self._gts, self._dts = defaultdict(list), defaultdict(list)
        for g in gts:
            g['ignore'] = bool(g.get('iscrowd', 0))
            if p.iou_type == 'keypoints':
                g['ignore'] = (g['num_keypoints'] == 0) or g['ignore']
            self._gts[g['image_id'], g['category_id']].append(g)
        for d in dts:
            self._dts[d['image_id'], d['category_id']].append(d)

    def _pair(self, img_id, cat_id):
        p = self.params
        if p.use_cats:
            return self._gts[img_id, cat_id], self._dts[img_id, cat_id]
        return ([g for c in p.cat_ids for g in self._gts[img_id, c]], [d for c in p.cat_ids for d in self._dts[img_id, c]])

    def _similarity(self, img_id, cat_id):
        gts, dts = self._pair(img_id, cat_id)
        if not gts or not dts:
            return np.zeros((0, 0))
        order = np.argsort([-d['score'] for d in dts], kind='mergesort')[:self.params.max_dets[-1]]
        dts = [dts[i] for i in order]
        if self.params.iou_type == 'bbox':
            return box_iou_xywh([d['bbox'] for d in dts], [g['bbox'] for g in gts], [int(g['iscrowd']) for g in gts])
        return oks([d['keypoints'] for d in dts], gts, landmark_meta()['oks_sigmas'])

    def _match(self, img_id, cat_id, sim, area, max_det):
        p = self.params
        gts, dts = self._pair(img_id, cat_id)
        if not gts and not dts:
            return None
        g_ign = np.array([int(g['ignore'] or g['area'] < area[0] or g['area'] > area[1]) for g in gts], dtype=np.int64)
        g_order = np.argsort(g_ign, kind='mergesort')
        gts = [gts[i] for i in g_order]
        g_ign = g_ign[g_order]
        d_order = np.argsort([-d['score'] for d in dts], kind='mergesort')[:max_det]
        dts = [dts[i] for i in d_order]
        crowd = [int(g['iscrowd']) for g in gts]
        sim = sim[:, g_order] if sim.size else sim
        T, G, D = len(p.iou_thrs), len(gts), len(dts)
        g_match, d_match, d_ign = np.zeros((T, G)), np.zeros((T, D)), np.zeros((T, D), dtype=bool)
        if sim.size:
            for ti, thr in enumerate(p.iou_thrs):
                for di in range(D):
                    best, m = min(thr, 1 - 1e-10), -1
                    for gi in range(G):
                        if g_match[ti, gi] > 0 and not crowd[gi]:
                            continue
                        if m > -1 and g_ign[m] == 0 and g_ign[gi] == 1:
                            break      # a regular match exists and only ignore regions follow
                        if sim[di, gi] < best:
                            continue
                        best, m = sim[di, gi], gi
                    if m >= 0:
                        d_ign[ti, di] = bool(g_ign[m])
                        d_match[ti, di] = gts[m]['id']
                        g_match[ti, m] = dts[di]['id']
        outside = np.array([d['area'] < area[0] or d['area'] > area[1] for d in dts], dtype=bool).reshape(1, D)
        d_ign = d_ign | ((d_match == 0) & outside)
        return dict(d_match=d_match, d_scores=np.array([d['score'] for d in dts], dtype=np.float64), g_ignore=g_ign,
                    d_ignore=d_ign)

    def evaluate(self):
        p = self.params
        p.img_ids = list(np.unique(p.img_ids))
        if p.use_cats:
            p.cat_ids = list(np.unique(p.cat_ids))
        p.max_dets = sorted(p.max_dets)
        self._prepare()
        cats = p.cat_ids if p.use_cats else [-1]
        sims = {(i, c): self._similarity(i, c) for i in p.img_ids for c in cats}
        top = p.max_dets[-1]
        self.eval_imgs = [self._match(i, c, sims[i, c], a, top) for c in cats for a in p.area_rng for i in p.img_ids]
        return self

    def accumulate(self):
        p = self.params
        cats = p.cat_ids if p.use_cats else [-1]
        T, R, K, A, M, I = len(p.iou_thrs), len(p.rec_thrs), len(cats), len(p.area_rng), len(p.max_dets), len(p.img_ids)
        precision, recall, scores = -np.ones((T, R, K, A, M)), -np.ones((T, K, A, M)), -np.ones((T, R, K, A, M))
        eps = np.spacing(1)
        for k in range(K):
            for a in range(A):
                cell = [e for e in self.eval_imgs[(k * A + a) * I:(k * A + a + 1) * I] if e is not None]
                if not cell:
                    continue
                g_ign = np.concatenate([e['g_ignore'] for e in cell])
                n_gt = int(np.count_nonzero(g_ign == 0))
                if n_gt == 0:
                    continue
                for m, max_det in enumerate(p.max_dets):
                    sc = np.concatenate([e['d_scores'][:max_det] for e in cell])
                    order = np.argsort(-sc, kind='mergesort')
                    sc = sc[order]
                    matched = np.concatenate([e['d_match'][:, :max_det] for e in cell], axis=1)[:, order] != 0
                    ignored = np.concatenate([e['d_ignore'][:, :max_det] for e in cell], axis=1)[:, order]
                    tp = np.cumsum(matched & ~ignored, axis=1).astype(np.float64)
                    fp = np.cumsum(~matched & ~ignored, axis=1).astype(np.float64)
                    for t in range(T):
                        rc = tp[t] / n_gt
                        pr = tp[t] / (fp[t] + tp[t] + eps)
                        recall[t, k, a, m] = rc[-1] if len(rc) else 0
                        pr = np.maximum.accumulate(pr[::-1])[::-1]      # precision envelope
                        pos = np.searchsorted(rc, p.rec_thrs, side='left')
                        ok = pos < len(pr)
                        q, s = np.zeros(R), np.zeros(R)
                        q[ok], s[ok] = pr[pos[ok]], sc[pos[ok]]
                        precision[t, :, k, a, m], scores[t, :, k, a, m] = q, s
        self.eval = dict(counts=[T, R, K, A, M], precision=precision, recall=recall, scores=scores)
        return self

    def _mean(self, ap, iou_thr=None, area='all', max_det=100):
        p = self.params
        a = [i for i, lbl in enumerate(p.area_lbl) if lbl == area]
        m = [i for i, d in enumerate(p.max_dets) if d == max_det]
        s = self.eval['precision' if ap else 'recall']
        if iou_thr is not None:
            s = s[np.where(iou_thr == p.iou_thrs)[0]]
        s = s[:, :, :, a, m] if ap else s[:, :, a, m]
        s = s[s > -1]
        return -1.0 if s.size == 0 else float(np.mean(s))

    def summarize(self, verbose=True):
        if not self.eval:
            raise RuntimeError('Please run accumulate() first')
        p = self.params
        if p.iou_type == 'bbox':
            big = p.max_dets[2]
            rows = [(1, None, 'all', big), (1, .5, 'all', big), (1, .75, 'all', big), (1, None, 'small', big),
                    (1, None, 'medium', big), (1, None, 'large', big), (0, None, 'all', p.max_dets[0]),
                    (0, None, 'all', p.max_dets[1]), (0, None, 'all', big), (0, None, 'small', big),
                    (0, None, 'medium', big), (0, None, 'large', big)]
            rows[0] = (1, None, 'all', 100)
        else:
            rows = [(ap, thr, area, 20) for ap in (1, 0)
                    for thr, area in ((None, 'all'), (.5, 'all'), (.75, 'all'), (None, 'medium'), (None, 'large'))]
        stats = []
        for ap, thr, area, md in rows:
            v = self._mean(ap, thr, area, md)
            stats.append(v)
            if verbose:
                span = '{:0.2f}:{:0.2f}'.format(p.iou_thrs[0], p.iou_thrs[-1]) if thr is None else '{:0.2f}'.format(thr)
                print(' {:<18} {} @[ IoU={:<9} | area={:>6s} | maxDets={:>3d} ] = {:0.3f}'.format(
                    'Average Precision' if ap else 'Average Recall', '(AP)' if ap else '(AR)', span, area, md, v))
        self.stats = np.array(stats)
        return self.stats


def coco_eval(result_files, result_types, coco, verbose=True):
    """``mmdet.core.coco_eval`` for the result types of this path ('bbox', 'keypoints'); returns {type: stats}."""
    if not isinstance(coco, CocoIndex):
        coco = CocoIndex(coco)
    out = {}
    for res_type in result_types:
        if res_type not in ('bbox', 'keypoints'):
            raise ValueError('unsupported result type {!r}'.format(res_type))
        ev = CocoEvaluator(coco, coco.load_results(result_files[res_type]), res_type)
        ev.params.img_ids = coco.get_img_ids()
        out[res_type] = ev.evaluate().accumulate().summarize(verbose)
    return out
