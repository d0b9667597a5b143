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
