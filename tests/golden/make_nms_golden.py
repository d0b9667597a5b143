"""Generate tests/golden/nms_golden.npz with the COMPILED REFERENCE kernels (run in the build
container only -- needs /root/reference):

    python tests/golden/make_nms_golden.py

Inputs are seeded synthetic boxes; expected outputs are whatever the unmodified reference
sources (nms_cpu.cpp, soft_nms_cpu.pyx, built by oracle/build_ref.py) return.  The .npz holds
only data: inputs, thresholds, kept indices, decayed boxes.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import build_ref  # noqa: E402


def make_boxes(rng, n, img_w=1333, img_h=800, cluster=True, quantize=False):
    """DeepFashion2-shaped detections: a few object clusters with jittered duplicates."""
    if n == 0:
        return np.zeros((0, 5), np.float32)
    if cluster:
        k = max(1, n // 12)
        cx = rng.uniform(100, img_w - 100, k)
        cy = rng.uniform(100, img_h - 100, k)
        w = rng.uniform(60, 600, k)
        h = rng.uniform(60, 600, k)
        which = rng.integers(0, k, n)
        jit = rng.normal(0, 12, (n, 4))
        x1 = cx[which] - w[which] / 2 + jit[:, 0]
        y1 = cy[which] - h[which] / 2 + jit[:, 1]
        x2 = cx[which] + w[which] / 2 + jit[:, 2]
        y2 = cy[which] + h[which] / 2 + jit[:, 3]
    else:
        x1 = rng.uniform(0, img_w - 50, n)
        y1 = rng.uniform(0, img_h - 50, n)
        x2 = x1 + rng.uniform(5, 400, n)
        y2 = y1 + rng.uniform(5, 400, n)
    x1 = np.clip(x1, 0, img_w); x2 = np.clip(x2, 0, img_w)
    y1 = np.clip(y1, 0, img_h); y2 = np.clip(y2, 0, img_h)
    x2 = np.maximum(x2, x1); y2 = np.maximum(y2, y1)
    s = rng.uniform(0.05, 1.0, n)
    d = np.stack([x1, y1, x2, y2, s], 1).astype(np.float32)
    if quantize:  # integer coordinates make exact IoU == thr ties reachable
        d[:, :4] = np.round(d[:, :4] / 8) * 8
    # distinct scores: the reference's sort order on ties is unspecified (nms_cpu.cpp:20)
    d[:, 4] = (np.argsort(np.argsort(d[:, 4])) + 1).astype(np.float32) / (n + 1)
    return d


def main():
    import torch
    ref = build_ref.load()
    assert ref is not None, 'reference sources not available'
    ref_nms, ref_soft = ref
    rng = np.random.default_rng(20260101)
    out = {}
    cases = []
    # hard NMS: sizes incl. empty, single, ragged, nms_pre-sized (1000) and serial-head sized (3350)
    for i, (n, thr, cluster, quant) in enumerate([
            (0, 0.5, True, False), (1, 0.5, True, False), (2, 0.5, True, False),
            (17, 0.5, True, False), (64, 0.5, True, False), (65, 0.3, False, False),
            (300, 0.5, True, False), (300, 0.5, True, True), (1000, 0.5, True, False),
            (1000, 0.7, False, False), (1000, 0.5, True, True), (3350, 0.5, True, False)]):
        d = make_boxes(rng, n, cluster=cluster, quantize=quant)
        keep = ref_nms.nms(torch.from_numpy(d), float(thr)).numpy().astype(np.int64)
        out['nms%d_dets' % i] = d
        out['nms%d_thr' % i] = np.float32(thr)
        out['nms%d_keep' % i] = keep
        cases.append(('nms', i, n, thr, len(keep)))
    # an exact IoU == thr tie: boxes [0,0,9,9] and [0,5,9,14] -> inter 50, union 150, IoU 1/3;
    # and IoU exactly 0.5: [0,0,9,9] vs [0,0,9,4]... areas 100 and 50, inter 50 -> 0.5
    tie = np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 4, 0.8], [20, 20, 29, 29, 0.7],
                    [20, 20, 29, 24, 0.95]], np.float32)
    out['nmstie_dets'] = tie
    out['nmstie_thr'] = np.float32(0.5)
    out['nmstie_keep'] = ref_nms.nms(torch.from_numpy(tie), 0.5).numpy().astype(np.int64)
    # soft-NMS
    for i, (n, thr, method, sigma, min_score) in enumerate([
            (0, 0.5, 1, 0.5, 0.001), (1, 0.5, 1, 0.5, 0.001), (40, 0.5, 1, 0.5, 0.05),
            (40, 0.5, 2, 0.5, 0.05), (300, 0.5, 1, 0.5, 0.05), (300, 0.3, 2, 0.3, 0.01),
            (1000, 0.5, 1, 0.5, 0.05), (1000, 0.5, 2, 0.5, 0.001), (3350, 0.5, 1, 0.5, 0.05)]):
        d = make_boxes(rng, n)
        if n == 0:
            nd, ni = d.copy(), np.zeros(0, np.int64)
        else:
            nd, ni = ref_soft(d, float(thr), method=method, sigma=float(sigma),
                              min_score=float(min_score))
        out['soft%d_dets' % i] = d
        out['soft%d_cfg' % i] = np.array([thr, method, sigma, min_score], np.float64)
        out['soft%d_new' % i] = np.asarray(nd, np.float32)
        out['soft%d_inds' % i] = np.asarray(ni, np.int64)
        cases.append(('soft', i, n, thr, len(ni)))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'nms_golden.npz')
    np.savez_compressed(path, **out)
    for c in cases:
        print(c)
    print('tie keep', out['nmstie_keep'])
    print('wrote', path, os.path.getsize(path), 'bytes')


def large():
    """tests/golden/nms_large_golden.npz: segments beyond the HIP kernel's on-chip limit of 4096 boxes (own seed, so
    nms_golden.npz is unaffected).  Only the seeds, thresholds and kept indices are stored: the boxes are re-created
    by make_boxes from the seed."""
    import torch
    ref_nms, _ = build_ref.load()
    out = {}
    for i, (n, thr, cluster, quant, seed) in enumerate([(4097, 0.5, True, False, 1), (8000, 0.5, True, False, 2),
                                                        (8000, 0.3, False, False, 3), (12000, 0.5, True, True, 4)]):
        d = make_boxes(np.random.default_rng(seed), n, cluster=cluster, quantize=quant)
        keep = ref_nms.nms(torch.from_numpy(d), float(thr)).numpy().astype(np.int64)
        out['case%d' % i] = np.array([n, seed, int(cluster), int(quant)], np.int64)
        out['thr%d' % i] = np.float32(thr)
        out['keep%d' % i] = keep
        out['checksum%d' % i] = np.float64(d.astype(np.float64).sum())
        print('large', i, n, thr, 'kept', len(keep))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'nms_large_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    if 'large' in sys.argv[1:]:
        large()
    else:
        main()
