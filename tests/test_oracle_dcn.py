"""Cross-check the deformable-conv / focal-loss oracle against independent formulations.

The reference has no CPU path and no tests for these ops ("parity unpinned"), so
the restatement in oracle/ is validated three ways: zero offsets == F.conv2d; arbitrary offsets
== F.grid_sample + einsum (fp64, 1e-10); all gradients == autograd of that formulation.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
from tests import torch_ref

CASES = [
    # N, C, H, W, O, k, stride, pad, dil, groups, dg
    (2, 8, 9, 11, 6, 3, 1, 1, 1, 1, 1),
    (1, 4, 7, 6, 4, 5, 1, 2, 1, 1, 1),
    (2, 4, 8, 8, 4, 7, 1, 3, 1, 1, 1),
    (2, 8, 10, 9, 8, 3, 2, 1, 1, 2, 1),
    (1, 8, 9, 9, 4, 3, 1, 2, 2, 1, 2),
    (3, 12, 6, 7, 6, 3, 1, 1, 1, 3, 4),
]


def _make(case, dtype, seed=0, offset_scale=2.0, with_mask=False):
    N, C, H, W, O, k, s, p, d, g, dg = case
    rng = np.random.default_rng(seed)
    Ho, Wo = oracle.conv_output_size(H, W, k, k, s, p, d)
    x = rng.normal(size=(N, C, H, W)).astype(dtype)
    off = (rng.normal(size=(N, dg * 2 * k * k, Ho, Wo)) * offset_scale).astype(dtype)
    w = (rng.normal(size=(O, C // g, k, k)) * 0.1).astype(dtype)
    go = rng.normal(size=(N, O, Ho, Wo)).astype(dtype)
    mask = rng.uniform(size=(N, dg * k * k, Ho, Wo)).astype(dtype) if with_mask else None
    return x, off, w, go, mask


@pytest.mark.parametrize('case', CASES)
def test_zero_offset_is_plain_conv(case):
    N, C, H, W, O, k, s, p, d, g, dg = case
    x, off, w, _, _ = _make(case, np.float64)
    out = oracle.deform_conv_forward(x, off * 0, w, s, p, d, g, dg)
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), None, s, p, d, g).numpy()
    np.testing.assert_allclose(out, ref, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize('with_mask', [False, True])
@pytest.mark.parametrize('case', CASES)
def test_forward_matches_grid_sample(case, with_mask):
    N, C, H, W, O, k, s, p, d, g, dg = case
    x, off, w, _, mask = _make(case, np.float64, seed=1, with_mask=with_mask)
    bias = np.linspace(-1, 1, O) if with_mask else None
    out = oracle.deform_conv_forward(x, off, w, s, p, d, g, dg, mask=mask, bias=bias)
    ref = torch_ref.deform_conv(torch.from_numpy(x), torch.from_numpy(off), torch.from_numpy(w), s, p,
                                d, g, dg, mask=None if mask is None else torch.from_numpy(mask),
                                bias=None if bias is None else torch.from_numpy(bias)).numpy()
    np.testing.assert_allclose(out, ref, rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize('with_mask', [False, True])
@pytest.mark.parametrize('case', CASES)
def test_backward_matches_autograd(case, with_mask):
    N, C, H, W, O, k, s, p, d, g, dg = case
    x, off, w, go, mask = _make(case, np.float64, seed=2, with_mask=with_mask)
    tx, to, tw = (torch.from_numpy(a).requires_grad_() for a in (x, off, w))
    tm = torch.from_numpy(mask).requires_grad_() if with_mask else None
    out = torch_ref.deform_conv(tx, to, tw, s, p, d, g, dg, mask=tm)
    out.backward(torch.from_numpy(go))
    res = oracle.deform_conv_backward(x, off, w, go, s, p, d, g, dg, mask=mask, with_bias=with_mask)
    np.testing.assert_allclose(res['grad_input'], tx.grad.numpy(), rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(res['grad_offset'], to.grad.numpy(), rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(res['grad_weight'], tw.grad.numpy(), rtol=1e-8, atol=1e-9)
    if with_mask:
        np.testing.assert_allclose(res['grad_mask'], tm.grad.numpy(), rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(res['grad_bias'], go.sum((0, 2, 3)), rtol=1e-12)


def test_float32_close_to_float64():
    case = (2, 16, 12, 13, 8, 5, 1, 2, 1, 1, 1)
    x, off, w, go, _ = _make(case, np.float64, seed=3)
    o64 = oracle.deform_conv_forward(x, off, w, 1, 2, 1)
    o32 = oracle.deform_conv_forward(x.astype(np.float32), off.astype(np.float32),
                                     w.astype(np.float32), 1, 2, 1)
    np.testing.assert_allclose(o32, o64, rtol=2e-4, atol=2e-5)


def test_out_of_range_taps_are_zero():
    # offsets that push every tap outside (-1, H) x (-1, W): output and all grads must be 0
    case = (1, 4, 5, 5, 3, 3, 1, 1, 1, 1, 1)
    x, off, w, go, _ = _make(case, np.float64, seed=4)
    far = off * 0 + 100.0
    assert np.all(oracle.deform_conv_forward(x, far, w, 1, 1, 1) == 0)
    res = oracle.deform_conv_backward(x, far, w, go, 1, 1, 1)
    assert np.all(res['grad_input'] == 0) and np.all(res['grad_offset'] == 0)
    assert np.all(res['grad_weight'] == 0)


def test_kgdet_offset_convention():
    """KGDet passes offset = reppoint - base_grid (KP3:137), so tap t samples at pixel + reppoint
    (SURVEY 9.1).  With reppoint == 0 every tap reads the centre pixel: out = sum_t W_t x."""
    N, C, H, W, O, k = 1, 3, 6, 7, 2, 3
    rng = np.random.default_rng(5)
    x = rng.normal(size=(N, C, H, W))
    w = rng.normal(size=(O, C, k, k))
    base = np.stack(np.meshgrid(np.arange(-1, 2), np.arange(-1, 2), indexing='ij'), -1).reshape(-1)  # (y,x) row-major
    off = np.broadcast_to(-base.reshape(1, 2 * k * k, 1, 1).astype(np.float64), (N, 2 * k * k, H, W))
    out = oracle.deform_conv_forward(x, np.ascontiguousarray(off), w, 1, 1, 1)
    ref = np.einsum('oc,nchw->nohw', w.sum((2, 3)), x)
    np.testing.assert_allclose(out, ref, rtol=1e-12, atol=1e-12)


# ----------------------------------------------------------------------------------------------
def test_focal_forward_backward_vs_reference_formula():
    rng = np.random.default_rng(0)
    n, c = 400, 13
    logits = (rng.normal(size=(n, c)) * 3).astype(np.float32)
    target = rng.integers(0, c + 1, n)
    target[:50] = 0
    tl = torch.from_numpy(logits).double().requires_grad_()
    ref = torch_ref.py_sigmoid_focal_loss(tl, torch.from_numpy(target), 2.0, 0.25)
    loss = oracle.sigmoid_focal_loss_forward(logits, target, 2.0, 0.25)
    np.testing.assert_allclose(loss, ref.detach().numpy(), rtol=2e-5, atol=1e-7)
    dl = rng.normal(size=(n, c)).astype(np.float32)
    ref.backward(torch.from_numpy(dl).double())
    grad = oracle.sigmoid_focal_loss_backward(logits, target, dl, 2.0, 0.25)
    np.testing.assert_allclose(grad, tl.grad.numpy(), rtol=5e-5, atol=1e-6)


def test_focal_extreme_logits_finite():
    logits = np.array([[-100.0, 100.0, 0.0], [88.0, -88.0, 30.0]], np.float32)
    target = np.array([2, 0])
    loss = oracle.sigmoid_focal_loss_forward(logits, target)
    assert np.all(np.isfinite(loss))
    g = oracle.sigmoid_focal_loss_backward(logits, target, np.ones_like(logits))
    assert np.all(np.isfinite(g))


@pytest.mark.parametrize('no_trans,group_size,part', [(True, 1, None), (False, 1, None), (False, 3, 3), (False, 1, 2)])
def test_psroi_oracle_matches_independent_formulation(no_trans, group_size, part):
    """oracle/psroi_oracle.inc (the restatement of deform_pool_cuda_kernel.cu:53-263, parity unpinned by the
    reference) against a dense grid_sample formulation; backward against its autograd, in fp64."""
    import torch
    from tests import torch_ref
    rng = np.random.default_rng(4)
    B, H, W, P, out_c = 2, 11, 13, 3, 4
    C = out_c * group_size * group_size
    part_size = P if part is None else part
    data = rng.normal(size=(B, C, H, W))
    R = 7
    x1 = rng.uniform(-20, 150, R); y1 = rng.uniform(-20, 120, R)
    rois = np.stack([rng.integers(0, B, R), x1, y1, x1 + rng.uniform(1, 120, R), y1 + rng.uniform(1, 120, R)], 1)
    rois[0, 1:] = [5, 5, 5, 5]            # degenerate RoI -> minimum size 0.1
    rois[1, 1:] = [300, 300, 400, 400]    # entirely outside -> every sample skipped, count 0
    rois[2, 1:] = [24, 18, 62, 55]        # well inside the 13 x 11 map (x 8): no sample is clamped
    rois[3, 1:] = [30, 22, 70, 60]
    offset = rng.normal(size=(R, 4, part_size, part_size)) * 0.5
    go = rng.normal(size=(R, out_c, P, P))
    scale, tstd = 1 / 8., 0.1
    ro, rc = oracle.deform_psroi_forward(data, rois, offset, scale, P, out_c, no_trans, group_size, part_size, 3, tstd)
    td = torch.from_numpy(data).requires_grad_()
    to = torch.from_numpy(offset).requires_grad_()
    out, cnt = torch_ref.deform_psroi_pool(td, torch.from_numpy(rois), to, scale, P, out_c, no_trans, group_size,
                                           part_size, 3, tstd)
    assert np.array_equal(cnt.numpy(), rc) and rc[1].max() == 0 and rc.max() == 9
    np.testing.assert_allclose(out.detach().numpy(), ro, rtol=0, atol=1e-12)
    out.backward(torch.from_numpy(go))
    gd, gt = oracle.deform_psroi_backward(go, rc, data, rois, offset, scale, P, out_c, no_trans, group_size, part_size,
                                          3, tstd)
    np.testing.assert_allclose(td.grad.numpy(), gd, rtol=0, atol=1e-12)
    if not no_trans:
        # the analytic offset gradient uses floor/ceil corners; autograd of the clamped coordinate has zero slope where
        # the clamp is active, the kernel does not clamp the derivative: compare where no sample was clamped
        interior = [r for r in range(R) if _roi_strictly_inside(rois[r], offset[r], scale, tstd, H, W)]
        assert len(interior) >= 2
        np.testing.assert_allclose(to.grad.numpy()[interior], gt[interior], rtol=0, atol=1e-10)


def _roi_strictly_inside(roi, off, scale, tstd, H, W):
    """every sample of this RoI lands in [0, W-1] x [0, H-1] whatever its (bounded) offsets: no clamping involved"""
    rsw, rsh = np.round(roi[1]) * scale - 0.5, np.round(roi[2]) * scale - 0.5
    rew, reh = (np.round(roi[3]) + 1) * scale - 0.5, (np.round(roi[4]) + 1) * scale - 0.5
    rw, rh = max(rew - rsw, 0.1), max(reh - rsh, 0.1)
    m = np.abs(off).max() * tstd
    return rsw - m * rw > 0 and rew + m * rw < W - 1 and rsh - m * rh > 0 and reh + m * rh < H - 1
