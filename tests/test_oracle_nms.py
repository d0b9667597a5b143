"""Pin the NMS / soft-NMS oracle to the reference's own CPU kernels.

Golden vectors in tests/golden/nms_golden.npz were produced by the compiled, unmodified
reference sources (tests/golden/make_nms_golden.py).  Bar: bit-exact indices and scores.
"""
import os

import numpy as np
import pytest

import oracle
from oracle import build_ref


@pytest.fixture(scope='module')
def G(golden_dir):
    return np.load(os.path.join(golden_dir, 'nms_golden.npz'))


def _cases(prefix, G):
    i = 0
    while '%s%d_dets' % (prefix, i) in G:
        yield i
        i += 1


def test_nms_matches_reference_golden(G):
    n_cases = 0
    for i in _cases('nms', G):
        keep = oracle.nms(G['nms%d_dets' % i], float(G['nms%d_thr' % i]))
        np.testing.assert_array_equal(keep, G['nms%d_keep' % i])
        n_cases += 1
    assert n_cases >= 12
    np.testing.assert_array_equal(oracle.nms(G['nmstie_dets'], float(G['nmstie_thr'])),
                                  G['nmstie_keep'])


def test_nms_ge_threshold_semantics():
    # IoU exactly 0.5 must suppress (nms_cpu.cpp:55 uses >=; the CUDA kernel uses >)
    d = np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 4, 0.8]], np.float32)
    assert oracle.nms(d, 0.5).tolist() == [0]
    assert oracle.nms(d, 0.5000001).tolist() == [0, 1]


def test_nms_empty_and_order():
    assert oracle.nms(np.zeros((0, 5), np.float32), 0.5).shape == (0,)
    # output is ascending box index, not score order (nms_cpu.cpp:58)
    d = np.array([[100, 100, 110, 110, 0.1], [0, 0, 10, 10, 0.9]], np.float32)
    assert oracle.nms(d, 0.5).tolist() == [0, 1]


def test_soft_nms_matches_reference_golden(G):
    names = {1: 'linear', 2: 'gaussian'}
    n_cases = 0
    for i in _cases('soft', G):
        thr, method, sigma, min_score = G['soft%d_cfg' % i]
        nd, ni = oracle.soft_nms(G['soft%d_dets' % i], float(thr), names[int(method)],
                                 float(sigma), float(min_score))
        np.testing.assert_array_equal(ni, G['soft%d_inds' % i])
        np.testing.assert_array_equal(nd, G['soft%d_new' % i])  # bit-exact float32
        n_cases += 1
    assert n_cases >= 9


def test_soft_nms_bad_method():
    with pytest.raises(ValueError):
        oracle.soft_nms(np.zeros((1, 5), np.float32), 0.5, method='nope')


@pytest.mark.skipif(not build_ref.available(), reason='reference sources not mounted')
def test_live_against_compiled_reference():
    """Fresh random inputs through the compiled reference ($KGDET_REF_BUILD, outside the tree) and the restatement."""
    import torch
    ref_nms, ref_soft = build_ref.load()
    rng = np.random.default_rng(7)
    for n in (5, 33, 257, 777):
        x1 = rng.uniform(0, 500, n); y1 = rng.uniform(0, 500, n)
        d = np.stack([x1, y1, x1 + rng.uniform(1, 300, n), y1 + rng.uniform(1, 300, n),
                      rng.permutation(n) / n + 0.01], 1).astype(np.float32)
        for thr in (0.3, 0.5, 0.8):
            np.testing.assert_array_equal(oracle.nms(d, thr),
                                          ref_nms.nms(torch.from_numpy(d), thr).numpy())
            for m, name in ((1, 'linear'), (2, 'gaussian')):
                rd, ri = ref_soft(d, thr, method=m, sigma=0.5, min_score=0.05)
                od, oi = oracle.soft_nms(d, thr, name, 0.5, 0.05)
                np.testing.assert_array_equal(oi, ri)
                np.testing.assert_array_equal(od, np.asarray(rd, np.float32))


def test_nms_large_segments_match_reference_golden(golden_dir):
    """segments beyond the HIP kernel's on-chip limit (4097 / 8000 / 12000 boxes): the oracle against the compiled
    reference's kept indices (tests/golden/nms_large_golden.npz, boxes re-created from their seeds)"""
    import os
    from tests.golden.make_nms_golden import make_boxes
    L = np.load(os.path.join(golden_dir, 'nms_large_golden.npz'))
    i = 0
    while 'case%d' % i in L:
        n, seed, cluster, quant = [int(v) for v in L['case%d' % i]]
        d = make_boxes(np.random.default_rng(seed), n, cluster=bool(cluster), quantize=bool(quant))
        assert float(d.astype(np.float64).sum()) == float(L['checksum%d' % i])     # the same boxes as the generator's
        np.testing.assert_array_equal(oracle.nms(d, float(L['thr%d' % i])), L['keep%d' % i])
        i += 1
    assert i == 4
