"""GPU parity: HIP deformable convolution (through the C ABI) vs the CPU oracle.

Tolerances: the default kernels multiply bf16 hi/lo splits of the fp32 operands (3 bf16 MFMAs per product, fp32
accumulate; 'exact' selects the f32-input MFMA kernel); the north-star bar is 1e-3 relative on coordinates.  Here outputs must agree with the float64 oracle to 2e-5 of the output scale and
with the float32 oracle (reference algorithm, different summation order) to the same bound.
"""
import os

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def _require_gpu():
    assert torch.cuda.is_available(), 'GPU tests need a GPU (no fallback)'


CASES = [
    # N, C, H, W, O, k, stride, pad, dil, groups, dg
    (2, 256, 25, 42, 256, 3, 1, 1, 1, 1, 1),    # KGDet 3x3
    (2, 256, 25, 42, 256, 5, 1, 2, 1, 1, 1),    # KGDet 5x5
    (2, 256, 25, 42, 256, 7, 1, 3, 1, 1, 1),    # KGDet 7x7
    (1, 256, 13, 17, 256, 3, 1, 1, 1, 1, 1),    # ragged pixel tile
    (3, 64, 20, 20, 128, 3, 1, 1, 1, 1, 1),     # O < 256 (padded rows)
    (2, 40, 11, 9, 24, 3, 2, 1, 1, 1, 1),       # C, O not multiples of the tile, stride 2
    (2, 32, 12, 12, 32, 3, 1, 2, 2, 2, 1),      # groups, dilation
    (2, 32, 10, 10, 16, 3, 1, 1, 1, 1, 4),      # deformable groups
    (1, 512, 8, 8, 512, 3, 1, 1, 1, 1, 1),      # two M tiles
    (1, 16, 5, 5, 8, 1, 1, 0, 1, 1, 1),         # 1x1 kernel, tiny
    (2, 256, 32, 40, 256, 3, 1, 1, 1, 1, 1),    # 1280 px: whole pixel tiles, above the gather fallback's 1088-px limit
    (2, 256, 50, 84, 256, 3, 1, 1, 1, 1, 1),    # config 5 (serial head), stride-16 level of 800x1344
    (2, 256, 100, 168, 256, 3, 1, 1, 1, 1, 1),  # config 5, stride-8 level: the 39.6 GFLOP call
]


# weight groups x deformable groups on the backward plane kernels (the calls below return "unsupported" instead of falling
# back, so a pass means the plane kernel computed the result)
GROUP_CASES = [
    (2, 32, 12, 12, 32, 3, 1, 2, 2, 2, 1),      # two weight groups
    (2, 32, 10, 10, 16, 3, 1, 1, 1, 2, 2),      # deformable group == weight group
    (2, 64, 10, 10, 32, 3, 1, 1, 1, 1, 4),      # deformable groups of 16 channels
    (2, 64, 12, 12, 64, 3, 1, 1, 1, 2, 4),      # two weight groups x four deformable groups
    (1, 96, 9, 11, 48, 5, 1, 2, 1, 3, 1),       # three weight groups
    (2, 256, 16, 20, 256, 3, 1, 1, 1, 1, 4),    # 64-channel runs: start inside a 256-row tile of the transposed weights
    (1, 512, 8, 8, 512, 3, 1, 1, 1, 1, 2),      # 256-channel runs: one weight tile each
    (1, 512, 8, 8, 256, 3, 1, 1, 1, 2, 2),      # two weight groups, each one deformable group of 256 channels
]


def _plane_map(case):
    """maps the LDS-plane kernels take directly (csrc/dcn_api.hip kPlaneMaxHW); larger maps go through the autograd
    entry points, which route them themselves"""
    return case[2] * case[3] <= 1344


def _make(case, seed=0, with_mask=False):
    N, C, H, W, O, k, s, p, d, g, dg = case
    rng = np.random.default_rng(seed)
    Ho, Wo = oracle.conv_output_size(H, W, k, k, s, p, d)
    x = rng.normal(size=(N, C, H, W)).astype(np.float32)
    off = (rng.normal(size=(N, dg * 2 * k * k, Ho, Wo)) * 2.0).astype(np.float32)
    w = (rng.normal(size=(O, C // g, k, k)) * 0.05).astype(np.float32)
    go = rng.normal(size=(N, O, Ho, Wo)).astype(np.float32)
    mask = rng.uniform(size=(N, dg * k * k, Ho, Wo)).astype(np.float32) if with_mask else None
    return x, off, w, go, mask


def _close(actual, desired, tol=2e-5):
    scale = max(float(np.abs(desired).max()), 1e-6)
    err = float(np.abs(actual.astype(np.float64) - desired).max()) / scale
    assert err < tol, 'max error %.3e of output scale (tol %.1e)' % (err, tol)


# forward arithmetic modes (include/kgdet_hip.h KGDET_DCN_*) and the error each must stay under, as a
# fraction of the output scale: the bf16 hi/lo split and the exact-fp32 kernel share the fp32 bound;
# single-rounded bf16 operands (autocast inference) carry 2^-9 per operand.
PRECISIONS = [('split', 2e-5), ('exact', 2e-5), ('bf16', 1e-2)]


@pytest.mark.parametrize('prec,tol', PRECISIONS)
@pytest.mark.parametrize('case', CASES)
def test_forward_v1(case, prec, tol):
    _require_gpu()
    from kgdet_amd import dcn
    N, C, H, W, O, k, s, p, d, g, dg = case
    x, off, w, _, _ = _make(case)
    with dcn.forward_precision(prec):
        out = dcn.deform_conv(torch.from_numpy(x).cuda(), torch.from_numpy(off).cuda(),
                              torch.from_numpy(w).cuda(), s, p, d, g, dg)
    torch.cuda.synchronize()
    ref64 = oracle.deform_conv_forward(x.astype(np.float64), off.astype(np.float64), w.astype(np.float64),
                                       s, p, d, g, dg)
    assert out.shape == ref64.shape
    _close(out.cpu().numpy(), ref64, tol)
    ref32 = oracle.deform_conv_forward(x, off, w, s, p, d, g, dg)
    _close(out.cpu().numpy(), ref32.astype(np.float64), tol)


@pytest.mark.parametrize('prec,tol', PRECISIONS)
@pytest.mark.parametrize('case', CASES[:3] + CASES[5:8])
def test_forward_v2_mask_bias(case, prec, tol):
    _require_gpu()
    from kgdet_amd import dcn
    N, C, H, W, O, k, s, p, d, g, dg = case
    x, off, w, _, mask = _make(case, seed=1, with_mask=True)
    bias = np.linspace(-1, 1, O).astype(np.float32)
    with dcn.forward_precision(prec):
        out = dcn.modulated_deform_conv(torch.from_numpy(x).cuda(), torch.from_numpy(off).cuda(),
                                        torch.from_numpy(mask).cuda(), torch.from_numpy(w).cuda(),
                                        torch.from_numpy(bias).cuda(), s, p, d, g, dg)
    ref64 = oracle.deform_conv_forward(x.astype(np.float64), off.astype(np.float64), w.astype(np.float64),
                                       s, p, d, g, dg, mask=mask.astype(np.float64), bias=bias.astype(np.float64))
    _close(out.cpu().numpy(), ref64, tol)


def test_large_map_forward_runs_on_split_operands():
    """maps beyond the LDS plane (config 5's stride-8 / stride-16 levels): the default arithmetic takes the split-operand
    kernel with gathered corners (plane_role MODE 2), not the exact-fp32 kernel -- close to it, not bit-equal, repeatable"""
    _require_gpu()
    from kgdet_amd import dcn
    case = (2, 64, 40, 48, 64, 3, 1, 1, 1, 1, 1)
    N, C, H, W, O, k, s, p, d, g, dg = case
    x, off, w, _, _ = _make(case, seed=21)
    tx, to, tw = (torch.from_numpy(a).cuda() for a in (x, off, w))
    with torch.no_grad():
        with dcn.forward_precision('split'):
            a = dcn.deform_conv(tx, to, tw, s, p, d, g, dg)
            a2 = dcn.deform_conv(tx, to, tw, s, p, d, g, dg)
        with dcn.forward_precision('exact'):
            b = dcn.deform_conv(tx, to, tw, s, p, d, g, dg)
    assert torch.equal(a, a2)
    assert not torch.equal(a, b), 'the split-operand path was expected for the default arithmetic'
    _close(a.cpu().numpy(), b.double().cpu().numpy(), 2e-5)
    # two weight groups and a modulation mask on the same path
    case2 = (1, 64, 40, 48, 32, 3, 1, 1, 1, 2, 1)
    x, off, w, _, mask = _make(case2, seed=22, with_mask=True)
    out = dcn.modulated_deform_conv(*(torch.from_numpy(t).cuda() for t in (x, off, mask, w)), None, 1, 1, 1, 2, 1)
    ref = oracle.deform_conv_forward(x.astype(np.float64), off.astype(np.float64), w.astype(np.float64), 1, 1, 1, 2, 1,
                                     mask=mask.astype(np.float64))
    _close(out.detach().cpu().numpy(), ref, 2e-5)
    # four deformable groups of 16 channels
    case3 = (1, 64, 40, 48, 32, 3, 1, 1, 1, 1, 4)
    x, off, w, _, _ = _make(case3, seed=23)
    out = dcn.deform_conv(*(torch.from_numpy(t).cuda() for t in (x, off, w)), 1, 1, 1, 1, 4)
    ref = oracle.deform_conv_forward(x.astype(np.float64), off.astype(np.float64), w.astype(np.float64), 1, 1, 1, 1, 4)
    _close(out.detach().cpu().numpy(), ref, 2e-5)


def test_forward_deterministic_and_zero_offset_is_conv():
    _require_gpu()
    from kgdet_amd import dcn
    case = CASES[2]
    x, off, w, _, _ = _make(case, seed=3)
    tx, to, tw = (torch.from_numpy(a).cuda() for a in (x, off, w))
    a = dcn.deform_conv(tx, to, tw, 1, 3, 1)
    b = dcn.deform_conv(tx, to, tw, 1, 3, 1)
    assert torch.equal(a, b), 'split-K fix-up must be order-deterministic'
    z = dcn.deform_conv(tx, torch.zeros_like(to), tw, 1, 3, 1)
    ref = torch.nn.functional.conv2d(tx.double(), tw.double(), None, 1, 3).float()
    _close(z.cpu().numpy(), ref.cpu().numpy().astype(np.float64))


def test_errors_match_reference_behaviour():
    _require_gpu()
    from kgdet_amd import dcn
    x = torch.zeros(2, 16, 8, 8)
    with pytest.raises(NotImplementedError):      # deform_conv.py:44-45
        dcn.deform_conv(x, torch.zeros(2, 18, 8, 8), torch.zeros(8, 16, 3, 3), 1, 1, 1)
    xc = x.cuda()
    with pytest.raises(RuntimeError):             # shape_check: offset channels
        dcn.deform_conv(xc, torch.zeros(2, 10, 8, 8).cuda(), torch.zeros(8, 16, 3, 3).cuda(), 1, 1, 1)
    with pytest.raises(ValueError):               # deform_conv.py:26-29
        dcn.deform_conv(xc[0], torch.zeros(2, 18, 8, 8).cuda(), torch.zeros(8, 16, 3, 3).cuda(), 1, 1, 1)


BWD_CASES = [
    (2, 256, 25, 42, 256, 3, 1, 1, 1, 1, 1),
    (2, 256, 25, 42, 256, 7, 1, 3, 1, 1, 1),
    (1, 256, 13, 17, 256, 5, 1, 2, 1, 1, 1),
    (3, 64, 20, 20, 128, 3, 1, 1, 1, 1, 1),
    (2, 40, 11, 9, 24, 3, 2, 1, 1, 1, 1),
    (2, 32, 12, 12, 32, 3, 1, 2, 2, 2, 1),
    (2, 32, 10, 10, 16, 3, 1, 1, 1, 2, 2),      # deformable group == weight group
    (2, 64, 10, 10, 32, 3, 1, 1, 1, 1, 4),      # deformable groups of 16 channels (plane kernels: one channel run each)
    (2, 64, 12, 12, 64, 3, 1, 1, 1, 2, 4),      # two weight groups x four deformable groups
    (1, 96, 9, 11, 48, 5, 1, 2, 1, 3, 1),       # three weight groups
    (1, 512, 8, 8, 512, 3, 1, 1, 1, 1, 1),      # two channel tiles -> atomic grad_offset
    (2, 256, 32, 40, 256, 3, 1, 1, 1, 1, 1),
    (2, 256, 50, 84, 256, 3, 1, 1, 1, 1, 1),    # config 5 large maps (reppoints_head_kp_serial.py:143-161)
    (2, 256, 100, 168, 256, 3, 1, 1, 1, 1, 1),
]


@pytest.mark.parametrize('with_mask', [False, True])
@pytest.mark.parametrize('case', [(1, 3, 40, 41, 18, 3, 1, 1, 1, 1, 1), (2, 6, 38, 40, 54, 3, 1, 1, 1, 2, 1)])
def test_large_map_backward_takes_any_channel_counts(case, with_mask):
    """ADVICE (round 5): on maps beyond the LDS plane the deterministic column-gradient kernels want 16-row output blocks and an
    even (tap x channel) count; output channels 18 / 27 per group or a 3-channel input with a 3x3 kernel raised since the
    float-atomic catch-all left.  They run zero-padded on the same kernels now (dcn._backward_input_padded): == oracle."""
    _require_gpu()
    from kgdet_amd import dcn
    N, C, H, W, O, k, s, p, d, g, dg = case
    x, off, w, go, mask = _make(case, seed=31, with_mask=with_mask)
    tx, to, tw = (torch.from_numpy(a).cuda().requires_grad_() for a in (x, off, w))
    if with_mask:
        tm = torch.from_numpy(mask).cuda().requires_grad_()
        out = dcn.modulated_deform_conv(tx, to, tm, tw, None, s, p, d, g, dg)
    else:
        out = dcn.deform_conv(tx, to, tw, s, p, d, g, dg)
    out.backward(torch.from_numpy(go).cuda())
    ref = oracle.deform_conv_backward(x.astype(np.float64), off.astype(np.float64), w.astype(np.float64), go.astype(np.float64),
                                      s, p, d, g, dg, mask=mask.astype(np.float64) if with_mask else None)
    _close(tx.grad.cpu().numpy(), ref['grad_input'], 5e-5)
    _close(to.grad.cpu().numpy(), ref['grad_offset'], 5e-5)
    _close(tw.grad.cpu().numpy(), ref['grad_weight'], 5e-5)
    if with_mask:
        _close(tm.grad.cpu().numpy(), ref['grad_mask'], 5e-5)


@pytest.mark.parametrize('case', BWD_CASES)
def test_backward_v1(case):
    _require_gpu()
    from kgdet_amd import dcn
    N, C, H, W, O, k, s, p, d, g, dg = case
    x, off, w, go, _ = _make(case, seed=5)
    tx, to, tw = (torch.from_numpy(a).cuda().requires_grad_() for a in (x, off, w))
    out = dcn.deform_conv(tx, to, tw, s, p, d, g, dg)
    out.backward(torch.from_numpy(go).cuda())
    torch.cuda.synchronize()
    ref = oracle.deform_conv_backward(x.astype(np.float64), off.astype(np.float64), w.astype(np.float64),
                                      go.astype(np.float64), s, p, d, g, dg)
    _close(tx.grad.cpu().numpy(), ref['grad_input'], 5e-5)
    _close(to.grad.cpu().numpy(), ref['grad_offset'], 5e-5)
    _close(tw.grad.cpu().numpy(), ref['grad_weight'], 5e-5)


LARGE_V2_CASE = (2, 64, 40, 48, 64, 3, 1, 1, 1, 1, 1)   # 1920 pixels: beyond the LDS-plane kernels, modulated (backbone DCNv2 maps)


@pytest.mark.parametrize('case', [BWD_CASES[0], BWD_CASES[4], BWD_CASES[5], BWD_CASES[6], BWD_CASES[7], BWD_CASES[8], LARGE_V2_CASE])
def test_backward_v2(case):
    _require_gpu()
    from kgdet_amd import dcn
    N, C, H, W, O, k, s, p, d, g, dg = case
    x, off, w, go, mask = _make(case, seed=6, with_mask=True)
    bias = np.linspace(-1, 1, O).astype(np.float32)
    tx, to, tm, tw, tb = (torch.from_numpy(a).cuda().requires_grad_() for a in (x, off, mask, w, bias))
    out = dcn.modulated_deform_conv(tx, to, tm, tw, tb, s, p, d, g, dg)
    out.backward(torch.from_numpy(go).cuda())
    ref = oracle.deform_conv_backward(x.astype(np.float64), off.astype(np.float64), w.astype(np.float64),
                                      go.astype(np.float64), s, p, d, g, dg, mask=mask.astype(np.float64),
                                      with_bias=True)
    _close(tx.grad.cpu().numpy(), ref['grad_input'], 5e-5)
    _close(to.grad.cpu().numpy(), ref['grad_offset'], 5e-5)
    _close(tm.grad.cpu().numpy(), ref['grad_mask'], 5e-5)
    _close(tw.grad.cpu().numpy(), ref['grad_weight'], 5e-5)
    _close(tb.grad.cpu().numpy(), ref['grad_bias'], 5e-5)


LARGE_GROUP_CASES = [
    # maps beyond the LDS plane (1344 pixels) with weight groups / deformable groups: channel runs on the column-gradient path
    (2, 32, 40, 40, 32, 3, 1, 1, 1, 2, 1),      # two weight groups
    (1, 64, 38, 42, 32, 3, 1, 1, 1, 1, 4),      # four deformable groups of 16 channels
    (2, 64, 40, 36, 64, 3, 1, 1, 1, 2, 4),      # two weight groups x four deformable groups
    (1, 96, 38, 38, 48, 3, 1, 1, 1, 3, 1),      # three weight groups
]


@pytest.mark.parametrize('with_mask', [False, True])
@pytest.mark.parametrize('case', LARGE_GROUP_CASES)
def test_large_map_backward_input_with_groups_has_no_atomics(case, with_mask):
    """Rounds 1-4 sent weight groups / deformable groups on maps beyond the LDS plane to a float-atomic scatter kernel (the
    reference's own non-determinism, deform_conv_cuda_kernel.cu:329).  Round 5: channel runs on the column-gradient path
    (csrc/dcn_backward_large.hip) -- grad_input, grad_offset (summed over the runs of a deformable group) and grad_mask against
    the float64 oracle, and bit-identical on a second call."""
    _require_gpu()
    from kgdet_amd import dcn
    N, C, H, W, O, k, s, p, d, g, dg = case
    x, off, w, go, mask = _make(case, seed=3, with_mask=with_mask)
    ref = oracle.deform_conv_backward(x.astype(np.float64), off.astype(np.float64), w.astype(np.float64), go.astype(np.float64),
                                      s, p, d, g, dg, mask=None if mask is None else mask.astype(np.float64))
    want = (ref['grad_input'], ref['grad_offset'], ref.get('grad_mask'))
    outs = []
    for _ in range(2):
        tx, to = torch.from_numpy(x).cuda().requires_grad_(), torch.from_numpy(off).cuda().requires_grad_()
        tw = torch.from_numpy(w).cuda().requires_grad_()
        if with_mask:
            tm = torch.from_numpy(mask).cuda().requires_grad_()
            y = dcn.modulated_deform_conv(tx, to, tm, tw, None, s, p, d, g, dg)
        else:
            tm = None
            y = dcn.deform_conv(tx, to, tw, s, p, d, g, dg)
        y.backward(torch.from_numpy(go).cuda())
        outs.append((tx.grad.clone(), to.grad.clone(), None if tm is None else tm.grad.clone()))
    _close(outs[0][0].cpu().numpy(), want[0], 5e-5)
    _close(outs[0][1].cpu().numpy(), want[1], 5e-5)
    if with_mask:
        _close(outs[0][2].cpu().numpy(), want[2], 5e-5)
    for a, b in zip(outs[0], outs[1]):
        assert a is None or torch.equal(a, b)


def test_large_map_v2_backward_is_repeatable():
    """modulated DCN on a map beyond the LDS-plane kernels: the column-gradient path (no float atomics) serves v2 too"""
    _require_gpu()
    from kgdet_amd import dcn
    N, C, H, W, O, k, s, p, d, g, dg = LARGE_V2_CASE
    x, off, w, go, mask = _make(LARGE_V2_CASE, seed=8, with_mask=True)
    grads = []
    for _ in range(2):
        tx, to, tm, tw = (torch.from_numpy(a).cuda().requires_grad_() for a in (x, off, mask, w))
        dcn.modulated_deform_conv(tx, to, tm, tw, None, s, p, d, g, dg).backward(torch.from_numpy(go).cuda())
        grads.append((tx.grad.clone(), to.grad.clone(), tm.grad.clone()))
    for a, b in zip(*grads):
        assert torch.equal(a, b)


@pytest.mark.parametrize('frac', [0.4, 1.0])
def test_large_map_backward_with_offsets_trained_onto_one_point(frac):
    """csrc/dcn_backward_large.hip inverse index: every (pixel, tap) of `frac` of the output pixels samples next to ONE point of a
    [1, 32, 40, 44] map -- the four cells around it collect thousands of entries each (0.4: ~6 300, sorted in LDS by
    large_cell_sort_long; 1.0: ~15 800, sorted in place in global memory), the lists the per-wave rank sort does not take.
    grad_input / grad_offset against the float64 oracle, and bit-repeatable."""
    _require_gpu()
    from kgdet_amd import dcn
    case = (1, 32, 40, 44, 32, 3, 1, 1, 1, 1, 1)
    N, C, H, W, O, k, s, p, d, g, dg = case
    x, off, w, go, _ = _make(case, seed=13)
    rng = np.random.default_rng(3)
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing='ij')
    hot = rng.random((H, W)) < frac
    for t in range(k * k):
        ty, tx_ = 20.3 + 0.05 * rng.standard_normal((H, W)), 17.6 + 0.05 * rng.standard_normal((H, W))
        base_y, base_x = ys - p + (t // k) * d, xs - p + (t % k) * d
        off[0, 2 * t] = np.where(hot, ty - base_y, off[0, 2 * t]).astype(np.float32)
        off[0, 2 * t + 1] = np.where(hot, tx_ - base_x, off[0, 2 * t + 1]).astype(np.float32)
    grads = []
    for _ in range(2):
        tx, to, tw = (torch.from_numpy(a).cuda().requires_grad_() for a in (x, off, w))
        dcn.deform_conv(tx, to, tw, s, p, d, g, dg).backward(torch.from_numpy(go).cuda())
        grads.append((tx.grad.clone(), to.grad.clone()))
    ref = oracle.deform_conv_backward(x.astype(np.float64), off.astype(np.float64), w.astype(np.float64),
                                      go.astype(np.float64), s, p, d, g, dg)
    _close(grads[0][0].cpu().numpy(), ref['grad_input'], 5e-5)
    _close(grads[0][1].cpu().numpy(), ref['grad_offset'], 5e-5)
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])


LARGE_WGRAD_CASES = [
    (2, 64, 40, 48, 64, 3, 1, 1, 1, 1, 1),      # one channel run
    (1, 64, 40, 48, 32, 3, 1, 1, 1, 2, 4),      # two weight groups x four deformable groups of 16 channels
    (1, 48, 38, 37, 40, 5, 1, 2, 1, 1, 1),      # 5x5 (four tap groups, the last one padded), ragged last pixel stage
    (1, 32, 80, 40, 24, 3, 2, 1, 1, 1, 1),      # stride 2: a large input map, a small output map
]


@pytest.mark.parametrize('with_mask', [False, True])
@pytest.mark.parametrize('case', LARGE_WGRAD_CASES)
def test_large_map_grad_weight_gathers_from_pixel_major_copy(case, with_mask):
    """maps beyond the LDS plane: grad_weight on split operands, corners gathered from the pixel-major copy of x
    (csrc/dcn_backward_weight_plane.hip dcn_bwd_weight_gather) -- against the f64 oracle, the fp32 kernel and itself"""
    _require_gpu()
    from kgdet_amd import dcn, _lib
    N, C, H, W, O, k, s, p, d, g, dg = case
    assert H * W > 1344
    x, off, w, go, mask = _make(case, seed=9, with_mask=with_mask)
    tx, to, tw, tg = (torch.from_numpy(a).cuda() for a in (x, off, w, go))
    tm = torch.from_numpy(mask).cuda() if with_mask else None
    shape = dcn._shape(tx, tw, (s, s), (p, p), (d, d), g, dg)
    needs = dict(input=False, offset=False, mask=False, weight=True, bias=False)
    got = [dcn._backward(tx, to, tm, tw, None, tg, shape, None, needs)[3] for _ in range(2)]
    assert torch.equal(got[0], got[1])            # slabs added in a fixed order
    try:
        _lib.check(_lib.lib().kgdet_set_option(0, 1), 'kgdet_set_option')
        exact = dcn._backward(tx, to, tm, tw, None, tg, shape, None, needs)[3]
    except NotImplementedError:     # the fp32 kernel's channel tiles cannot straddle deformable groups (case 1)
        assert g > 1 and dg > 1
        exact = None
    finally:
        _lib.check(_lib.lib().kgdet_set_option(0, 0), 'kgdet_set_option')
    ref = oracle.deform_conv_backward(x.astype(np.float64), off.astype(np.float64), w.astype(np.float64),
                                      go.astype(np.float64), s, p, d, g, dg,
                                      mask=mask.astype(np.float64) if with_mask else None)['grad_weight']
    _close(got[0].cpu().numpy(), ref, 2e-5)
    if exact is not None:
        _close(exact.cpu().numpy(), ref, 2e-5)


def test_grad_weight_deterministic():
    _require_gpu()
    from kgdet_amd import dcn
    x, off, w, go, _ = _make(BWD_CASES[1], seed=7)
    grads = []
    for _ in range(2):
        tx, to, tw = (torch.from_numpy(a).cuda().requires_grad_() for a in (x, off, w))
        dcn.deform_conv(tx, to, tw, 1, 3, 1).backward(torch.from_numpy(go).cuda())
        grads.append((to.grad.clone(), tw.grad.clone()))
    assert torch.equal(grads[0][0], grads[1][0])   # grad_offset: no atomics on this shape
    assert torch.equal(grads[0][1], grads[1][1])   # grad_weight: slab fix-up in fixed order


@pytest.mark.parametrize('B', [2, 8])
def test_grouped_forward_matches_single_calls(B):
    """kgdet_deform_conv_forward_grouped == n separate calls, bit for bit (same kernels, same split)... the
    schedule differs (B = 2: one static range per workgroup; B = 8: a four-round static schedule; the single calls:
    stream-K or other part counts), so fp32 summation order may differ: compare to fp32 round-off of the scale."""
    _require_gpu()
    from kgdet_amd import dcn
    torch.manual_seed(5)
    C, H, W = 256, 25, 42
    xs = [torch.randn(B, C, H, W, device='cuda') for _ in range(2)]
    ks = (3, 5, 7)
    offsets = [torch.randn(B, 2 * k * k, H, W, device='cuda') * 2 for k in ks]
    weights = [[torch.randn(64, C, k, k, device='cuda') * 0.05 for k in ks] for _ in xs]
    pads = [k // 2 for k in ks]
    outs = dcn.deform_conv_cat_multi(xs, offsets, weights, pads, relu=True)
    again = dcn.deform_conv_cat_multi(xs, offsets, weights, pads, relu=True)
    for i, x in enumerate(xs):
        assert torch.equal(outs[i], again[i]), 'grouped launch must be deterministic'
        ref = torch.cat([dcn.deform_conv(x, offsets[k], weights[i][k], 1, pads[k]) for k in range(3)], 1).relu()
        _close(outs[i].cpu().numpy(), ref.double().cpu().numpy(), 2e-6)
    # gradients flow to every input of the grouped function (no ReLU here: the two paths split the reduction
    # differently, and an output within round-off of zero may land on either side of the ReLU mask)
    for t in xs + offsets + [w for ws in weights for w in ws]:
        t.requires_grad_(True)
    outs = dcn.deform_conv_cat_multi(xs, offsets, weights, pads, relu=False)
    (outs[0].sum() + 2 * outs[1].sum()).backward()
    g_off = [o.grad.clone() for o in offsets]
    for t in xs + offsets + [w for ws in weights for w in ws]:
        assert t.grad is not None and torch.isfinite(t.grad).all()
        t.grad = None
    (dcn.deform_conv_cat(xs[0], offsets, weights[0], pads, relu=False).sum()
     + 2 * dcn.deform_conv_cat(xs[1], offsets, weights[1], pads, relu=False).sum()).backward()
    for a, o in zip(g_off, offsets):
        _close(a.cpu().numpy(), o.grad.double().cpu().numpy(), 1e-5)


@pytest.mark.parametrize('B,prec,tol', [(8, 'bf16', 1e-2), (8, 'split', 2e-5), (2, 'split', 2e-5)])
def test_grouped_head_stage_launch_against_the_oracle(B, prec, tol):
    """BASELINE config 2's exact launch (and the bench's roofline launch): the grouped head-stage forward -- 2 maps x 3x3 / 5x5 /
    7x7 at FULL width (256 -> 256 channels) on [B, 256, 25, 42], ReLU and the three channel windows of one buffer per map, in the
    arithmetic the inference batch uses (bf16 operands, B = 8) and in the training arithmetic -- compared DIRECTLY with the float64
    oracle (deform_conv_cuda.cpp:151-258 restated), not only with single calls of the same kernels."""
    _require_gpu()
    from kgdet_amd import dcn
    g = torch.Generator().manual_seed(11)
    C, H, W = 256, 25, 42
    xs = [torch.randn(B, C, H, W, generator=g) for _ in range(2)]
    ks = (3, 5, 7)
    offsets = [torch.randn(B, 2 * k * k, H, W, generator=g) * 2 for k in ks]
    weights = [[torch.randn(C, C, k, k, generator=g) * 0.02 for k in ks] for _ in xs]
    pads = [k // 2 for k in ks]
    with torch.no_grad(), dcn.forward_precision(prec):
        outs = dcn.deform_conv_cat_multi([x.cuda() for x in xs], [o.cuda() for o in offsets],
                                         [[w.cuda() for w in wl] for wl in weights], pads, relu=True)
    for i in range(2):
        got = outs[i].cpu().numpy()
        assert got.shape == (B, 3 * C, H, W)
        for k in range(3):
            # (images 0 and B - 1: the first and the last pixel tiles of the launch; the oracle's im2col + GEMM in float64)
            for b in (0, B - 1):
                ref = oracle.deform_conv_forward(xs[i][b:b + 1].numpy().astype(np.float64), offsets[k][b:b + 1].numpy().astype(np.float64),
                                                 weights[i][k].numpy().astype(np.float64), 1, pads[k], 1)
                _close(got[b:b + 1, k * C:(k + 1) * C], np.maximum(ref, 0.0), tol)


def test_autocast_contract():
    """under torch.autocast(bfloat16) the ops cast activations up to float32, use bf16 products
    (KGDET_DCN_BF16) and return float32; a non-float32 tensor outside autocast is rejected, never
    reinterpreted."""
    _require_gpu()
    from kgdet_amd import dcn
    case = CASES[0]
    x, off, w, _, _ = _make(case, seed=7)
    tx, to, tw = (torch.from_numpy(a).cuda() for a in (x, off, w))
    ref = dcn.deform_conv(tx, to, tw, 1, 1, 1)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out = dcn.deform_conv(tx.bfloat16(), to, tw, 1, 1, 1)
        cat = dcn.deform_conv_cat(tx.bfloat16(), [to], [tw], [1], relu=False)
    assert out.dtype == torch.float32 and cat.dtype == torch.float32
    ref_b = dcn.deform_conv(tx.bfloat16().float(), to, tw, 1, 1, 1)
    _close(out.cpu().numpy(), ref_b.double().cpu().numpy(), 1e-2)
    _close(cat.cpu().numpy(), ref_b.double().cpu().numpy(), 1e-2)
    assert not torch.equal(out, ref_b), 'bf16 autocast is expected to select the bf16-operand kernel'
    with pytest.raises(TypeError):
        dcn.deform_conv(tx.bfloat16(), to, tw, 1, 1, 1)
    assert torch.equal(ref, dcn.deform_conv(tx, to, tw, 1, 1, 1))


@pytest.mark.parametrize('case', [c for c in CASES if c[10] == 1 and _plane_map(c)] + GROUP_CASES)
def test_grad_input_plane_kernel(case):
    """kgdet_deform_conv_grad_input (transposed sampling on the plane kernel) vs the float64 oracle."""
    _require_gpu()
    from kgdet_amd import dcn
    N, C, H, W, O, k, s, p, d, g, dg = case
    x, off, w, go, _ = _make(case, seed=11)
    tx, to, tw, tg = (torch.from_numpy(a).cuda() for a in (x, off, w, go))
    shape = dcn._shape(tx, tw, (s, s), (p, p), (d, d), g, dg)
    gi = dcn.grad_input_plane(tx.shape, to, None, tw, tg, shape)
    gi2 = dcn.grad_input_plane(tx.shape, to, None, tw, tg, shape)
    assert torch.equal(gi, gi2), 'grad_input must be deterministic'
    ref = oracle.deform_conv_backward(x.astype(np.float64), off.astype(np.float64), w.astype(np.float64),
                                      go.astype(np.float64), s, p, d, g, dg)['grad_input']
    _close(gi.cpu().numpy(), ref, 5e-5)


def test_grad_input_plane_kernel_long_lists():
    """all taps of all pixels sample (nearly) the same spot: contribution lists of hundreds of entries exercise the
    overflow slots and the spill list of the inverse records"""
    _require_gpu()
    from kgdet_amd import dcn
    case = (1, 32, 12, 14, 48, 3, 1, 1, 1, 1, 1)
    N, C, H, W, O, k, s, p, d, g, dg = case
    x, off, w, go, _ = _make(case, seed=12)
    Ho, Wo = oracle.conv_output_size(H, W, k, k, s, p, d)
    ys, xs = np.meshgrid(np.arange(Ho), np.arange(Wo), indexing='ij')
    for t in range(k * k):   # cancel the regular grid: every sample lands near (5.3, 6.6)
        off[:, 2 * t] = 5.3 - (ys - p + t // k) + 0.01 * off[:, 2 * t]
        off[:, 2 * t + 1] = 6.6 - (xs - p + t % k) + 0.01 * off[:, 2 * t + 1]
    tx, to, tw, tg = (torch.from_numpy(a).cuda() for a in (x, off, w, go))
    shape = dcn._shape(tx, tw, (s, s), (p, p), (d, d), g, dg)
    gi = dcn.grad_input_plane(tx.shape, to, None, tw, tg, shape)
    ref = oracle.deform_conv_backward(x.astype(np.float64), off.astype(np.float64), w.astype(np.float64),
                                      go.astype(np.float64), s, p, d, g, dg)['grad_input']
    _close(gi.cpu().numpy(), ref, 5e-5)


def _keypoint_offsets(case, off, points_per_image=2, jitter=0.3, seed=3):
    """the offsets of a TRAINED keypoint-guided head: tap t of every location of an image samples one of a few key points
    (absolute positions), whatever the location -- `reppts - base grid` of KP3:135-143 once the points have converged"""
    N, C, H, W, O, k, s, p, d, g, dg = case
    rng = np.random.default_rng(seed)
    Ho, Wo = off.shape[2:]
    ys, xs = np.meshgrid(np.arange(Ho), np.arange(Wo), indexing='ij')
    off = off.copy()
    for b in range(N):
        for gi in range(dg):
            for t in range(k * k):
                ky = rng.uniform(1, H - 2, size=points_per_image)
                kx = rng.uniform(1, W - 2, size=points_per_image)
                which = (xs * points_per_image // Wo).clip(0, points_per_image - 1)     # left object / right object
                ch = gi * 2 * k * k + 2 * t
                off[b, ch] = ky[which] - (ys * s - p + (t // k) * d) + jitter * 0.1 * off[b, ch]
                off[b, ch + 1] = kx[which] - (xs * s - p + (t % k) * d) + jitter * 0.1 * off[b, ch + 1]
    return off


@pytest.mark.parametrize('case', [CASES[2], CASES[0], GROUP_CASES[3], GROUP_CASES[5]])
def test_grad_input_with_trained_head_offsets(case):
    """Round 4: with converged key points every (image, tap) has a few cells that collect HUNDREDS of contributions.  The
    grad_input path forms those cells' sums beforehand for all channels (dcn_inv_overflow_sums), so (a) the result still
    agrees with the float64 oracle, bit-repeatably, also across weight / deformable groups, and (b) the launch costs about
    what it costs with random offsets -- the per-cell list walk of rounds 2-3 took 6-7x longer on such offsets, which is
    what made the training step slow down from 12 to 16 ms within a few hundred steps."""
    _require_gpu()
    from kgdet_amd import dcn
    N, C, H, W, O, k, s, p, d, g, dg = case
    x, off, w, go, _ = _make(case, seed=21)
    offk = _keypoint_offsets(case, off)
    tx, tw, tg = (torch.from_numpy(a).cuda() for a in (x, w, go))
    to_rand, to_key = torch.from_numpy(off).cuda(), torch.from_numpy(offk).cuda()
    shape = dcn._shape(tx, tw, (s, s), (p, p), (d, d), g, dg)
    gi = dcn.grad_input_plane(tx.shape, to_key, None, tw, tg, shape)
    gi2 = dcn.grad_input_plane(tx.shape, to_key, None, tw, tg, shape)
    assert torch.equal(gi, gi2), 'grad_input must be deterministic'
    ref = oracle.deform_conv_backward(x.astype(np.float64), offk.astype(np.float64), w.astype(np.float64),
                                      go.astype(np.float64), s, p, d, g, dg)['grad_input']
    _close(gi.cpu().numpy(), ref, 5e-5)
    if C < 256:
        return

    packed = dcn.pack_weight(tw.contiguous(), shape)

    def timed(to):
        for _ in range(3):
            dcn.grad_input_plane(tx.shape, to, None, tw, tg, shape, packed)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dcn.grad_input_plane(tx.shape, to, None, tw, tg, shape, packed)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10
    t_rand, t_key = timed(to_rand), timed(to_key)
    # (measured 1.3-1.6x: the pre-aggregation reads every contribution's 1 KB row of grad_out once; rounds 2-3: 6-7x)
    # (round 6: the random-offset call got faster -- builders fused, 39 -> 24 us -- the trained one did not: its hot clusters' row reads
    # are what is left, profiles/r06_dcn_bwd_plane_kernels.md; the bound guards against the list walk coming back, not the ratio's decimals)
    assert t_key < 2.6 * t_rand, 'grad_input depends on the offset distribution again: %.3f ms vs %.3f ms' % (t_key, t_rand)


def _hot_columns():
    import ctypes
    from kgdet_amd import _lib
    buf = (ctypes.c_int * 64)()
    n = _lib.lib().kgdet_debug_dcn_hot_columns(buf, 64)
    assert n >= 0
    return [buf[i] for i in range(n)]


@pytest.mark.parametrize('H,W,O,points,expect', [
    (25, 42, 144, 2, 'hot'),       # the head's map; 144 output channels = one full 128-channel part + a 16-channel one
    (32, 41, 64, 2, 'hot'),        # 1312 pixels: two passes of the GEMM's 1152-pixel range
    (25, 42, 32, 2, 'over'),       # a column list of 256 (test switch): images with more hot cells take the cluster path whole
    (25, 42, 48, 0, 'none'),       # random offsets: no hot cells, the GEMM's workgroups find nothing to do
])
def test_grouped_backward_hot_cells_go_through_the_gemm(H, W, O, points, expect):
    """Round 6: in the grouped backward the cells that collect more than 64 contributions (converged key-point offsets) become columns
    of one dense MFMA GEMM per (problem, image) -- dcn_hot_gemm -- instead of the cluster path of dcn_inv_overflow_sums.  grad_input
    of the whole group (two maps x 3x3 / 5x5 / 7x7, summed per map by the fix-up) against the float64 oracle, bit-repeatably, with
    the test hook confirming which path the cells took."""
    _require_gpu()
    from kgdet_amd import dcn
    B, C = 2, 32
    rng = np.random.default_rng(17)
    ks = (3, 5, 7)
    xs = [rng.standard_normal((B, C, H, W)).astype(np.float32) for _ in range(2)]
    offs = [(rng.standard_normal((B, 2 * k * k, H, W)) * 2).astype(np.float32) for k in ks]
    if points:
        gy, gx = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
        which = (gx * points // W).clip(0, points - 1)
        for i, k in enumerate(ks):
            for b in range(B):
                for t in range(k * k):
                    ky, kx = rng.uniform(1, H - 2, size=points), rng.uniform(1, W - 2, size=points)
                    offs[i][b, 2 * t] = ky[which] - (gy - k // 2 + t // k) + 0.03 * offs[i][b, 2 * t]
                    offs[i][b, 2 * t + 1] = kx[which] - (gx - k // 2 + t % k) + 0.03 * offs[i][b, 2 * t + 1]
    ws = [[(rng.standard_normal((O, C, k, k)) * 0.05).astype(np.float32) for k in ks] for _ in xs]
    gos = [rng.standard_normal((B, 3 * O, H, W)).astype(np.float32) for _ in xs]

    def run():
        os.environ['KGDET_DCN_HOT_DEBUG'] = '1'
        if expect == 'over':
            os.environ['KGDET_DCN_HOT_MAX_COLS'] = '256'
        try:
            return run_()
        finally:
            del os.environ['KGDET_DCN_HOT_DEBUG']
            os.environ.pop('KGDET_DCN_HOT_MAX_COLS', None)

    def run_():
        txs = [torch.from_numpy(x).cuda().requires_grad_() for x in xs]
        tos = [torch.from_numpy(o).cuda().requires_grad_() for o in offs]
        tws = [[torch.from_numpy(w).cuda().requires_grad_() for w in wl] for wl in ws]
        outs = dcn.deform_conv_cat_multi(txs, tos, tws, [k // 2 for k in ks], relu=False)
        torch.autograd.backward(outs, [torch.from_numpy(g).cuda() for g in gos])
        return [t.grad.clone() for t in txs], _hot_columns()
    got, cols = run()
    again, _ = run()
    assert len(cols) == 3 * B, 'the grouped plane backward (and its hot-cell list) must have run: %r' % (cols,)
    if expect == 'none':
        assert max(cols) == 0
    elif expect == 'hot':
        assert min(cols) > 0 and max(cols) <= 4096
    else:
        assert max(cols) > 256 and min(cols) <= 256, cols       # (the 3x3 tensor's images fit, the 7x7's do not)
    for i in range(2):
        assert torch.equal(got[i], again[i]), 'grad_input must be deterministic'
        ref = np.zeros_like(xs[i], dtype=np.float64)
        for j, k in enumerate(ks):
            ref += oracle.deform_conv_backward(xs[i].astype(np.float64), offs[j].astype(np.float64), ws[i][j].astype(np.float64),
                                               gos[i][:, j * O:(j + 1) * O].astype(np.float64), 1, k // 2, 1, 1, 1)['grad_input']
        _close(got[i].cpu().numpy(), ref, 5e-5)


@pytest.mark.parametrize('case', [c for c in CASES if c[10] == 1 and c[9] == 1 and c[4] <= 256 and _plane_map(c)] +
                         [c for c in GROUP_CASES if (c[1] // c[9]) % (c[1] // c[10]) == 0 and c[4] // c[9] <= 256])   # a deformable group inside ONE weight group
def test_grad_offset_plane_kernel(case):
    """kgdet_deform_conv_grad_offset (column gradient in registers, feature plane in LDS) vs the float64 oracle."""
    _require_gpu()
    from kgdet_amd import dcn
    N, C, H, W, O, k, s, p, d, g, dg = case
    x, off, w, go, _ = _make(case, seed=13)
    tx, to, tw, tg = (torch.from_numpy(a).cuda() for a in (x, off, w, go))
    shape = dcn._shape(tx, tw, (s, s), (p, p), (d, d), g, dg)
    a = dcn.grad_offset_plane(tx, to, tw, tg, shape)
    b = dcn.grad_offset_plane(tx, to, tw, tg, shape)
    assert torch.equal(a, b), 'grad_offset must be deterministic'
    ref = oracle.deform_conv_backward(x.astype(np.float64), off.astype(np.float64), w.astype(np.float64),
                                      go.astype(np.float64), s, p, d, g, dg)['grad_offset']
    _close(a.cpu().numpy(), ref, 5e-5)


def test_weight_images_follow_fused_optimizer_updates():
    """fused optimizers change weights without bumping ``_version``: training must re-pack every call, and the
    inference cache must not survive a mode switch"""
    _require_gpu()
    from kgdet_amd import dcn
    torch.manual_seed(0)
    conv = dcn.DeformConv(32, 32, 3, padding=1).cuda()
    x = torch.randn(1, 32, 12, 12, device='cuda')
    off = torch.randn(1, 18, 12, 12, device='cuda')
    opt = torch.optim.Adam(conv.parameters(), lr=0.5, fused=True)
    y0 = conv(x, off)
    y0.sum().backward()
    v0 = conv.weight._version
    opt.step()
    assert conv.weight._version == v0, 'this torch build bumps _version in fused Adam: test premise gone'
    y1 = conv(x, off)
    assert not torch.allclose(y0, y1), 'forward after the optimizer step still used the old weight images'
    conv.eval()
    with torch.no_grad():
        a = conv(x, off)
        b = conv(x, off)        # served from the cache
    assert torch.equal(a, b) and torch.allclose(a, y1.detach())
    conv.train()
    conv(x, off).sum().backward()
    opt.step()
    conv.eval()
    with torch.no_grad():
        c = conv(x, off)
    assert not torch.allclose(a, c), 'inference after further training used stale weight images'


@pytest.mark.parametrize('case', [c for c in CASES if c[10] == 1 and c[9] == 1 and _plane_map(c)] + GROUP_CASES)
def test_grad_weight_plane_kernel(case):
    """kgdet_deform_conv_grad_weight_grouped (pixel-reduction GEMM on the plane kernel) vs the float64 oracle."""
    _require_gpu()
    from kgdet_amd import dcn
    N, C, H, W, O, k, s, p, d, g, dg = case
    x, off, w, go, _ = _make(case, seed=14)
    tx, to, tw, tg = (torch.from_numpy(a).cuda() for a in (x, off, w, go))
    shape = dcn._shape(tx, tw, (s, s), (p, p), (d, d), g, dg)
    a = dcn.grad_weights_grouped([tx], [to], [tg], [tw], [shape])
    b = dcn.grad_weights_grouped([tx], [to], [tg], [tw], [shape])
    assert a is not None and torch.equal(a[0], b[0]), 'grad_weight must be deterministic'
    ref = oracle.deform_conv_backward(x.astype(np.float64), off.astype(np.float64), w.astype(np.float64),
                                      go.astype(np.float64), s, p, d, g, dg)['grad_weight']
    _close(a[0].cpu().numpy(), ref, 5e-5)


@pytest.mark.parametrize('case', [CASES[0], CASES[2], CASES[5], GROUP_CASES[3]])
def test_grad_weight_stream_k_schedule_stays_correct(case):
    """KGDET_OPT_WGRAD_STREAMK: rounds 1-3's schedule (256 x 128 tiles dealt stream-K, partial tiles + fix-up) stays a working
    alternative to the output-stationary kernel the calls above run (round 4): both against the float64 oracle, and against
    each other to round-off"""
    _require_gpu()
    from kgdet_amd import _lib, dcn
    N, C, H, W, O, k, s, p, d, g, dg = case
    x, off, w, go, _ = _make(case, seed=15)
    tx, to, tw, tg = (torch.from_numpy(a).cuda() for a in (x, off, w, go))
    shape = dcn._shape(tx, tw, (s, s), (p, p), (d, d), g, dg)
    a = dcn.grad_weights_grouped([tx], [to], [tg], [tw], [shape])
    _lib.check(_lib.lib().kgdet_set_option(2, 1), 'kgdet_set_option')
    try:
        b = dcn.grad_weights_grouped([tx], [to], [tg], [tw], [shape])
    finally:
        _lib.check(_lib.lib().kgdet_set_option(2, 0), 'kgdet_set_option')
    ref = oracle.deform_conv_backward(x.astype(np.float64), off.astype(np.float64), w.astype(np.float64),
                                      go.astype(np.float64), s, p, d, g, dg)['grad_weight']
    _close(a[0].cpu().numpy(), ref, 5e-5)
    _close(b[0].cpu().numpy(), ref, 5e-5)
    _close(a[0].cpu().numpy(), b[0].double().cpu().numpy(), 2e-5)
    assert not torch.equal(a[0], b[0]) or a[0].numel() < 64, 'the option is expected to select the other schedule'


def test_more_tiles_than_slab_slots_take_the_safe_path():
    """a launch whose workgroup slices would meet more tile ranges than a workgroup has slab slots
    (here 192 images x 9 pixel tiles = 1728 tiles on 256 CUs, > 6 per workgroup) must not run the stream-K plane
    kernels (their slab index would run into the neighbour's slabs): results stay correct, forward and backward."""
    _require_gpu()
    from kgdet_amd import dcn
    case = (192, 16, 25, 42, 16, 3, 1, 1, 1, 1, 1)     # (a multiple of the default im2col_step = 64)
    N, C, H, W, O, k, s, p, d, g, dg = case
    x, off, w, go, _ = _make(case, seed=21)
    tx, to, tw = (torch.from_numpy(a).cuda().requires_grad_() for a in (x, off, w))
    out = dcn.deform_conv(tx, to, tw, s, p, d, g, dg)
    out.backward(torch.from_numpy(go).cuda())
    torch.cuda.synchronize()
    f64 = lambda a: a.astype(np.float64)
    _close(out.detach().cpu().numpy(), oracle.deform_conv_forward(f64(x), f64(off), f64(w), s, p, d, g, dg))
    ref = oracle.deform_conv_backward(f64(x), f64(off), f64(w), f64(go), s, p, d, g, dg)
    _close(tx.grad.cpu().numpy(), ref['grad_input'], 5e-5)
    _close(tw.grad.cpu().numpy(), ref['grad_weight'], 5e-5)
    # grad_offset: the derivative of the bilinear sample jumps where a sampling position crosses a cell boundary, and
    # among 3.6 million samples a few positions round onto the other side of a boundary in fp32 (the kernels', and
    # the reference's, arithmetic) than in the float64 oracle: the plane kernels and the exact-fp32 kernels give the
    # SAME values there (tools/exp_many_tiles.py).  So: all but <= 1e-5 of the elements within tolerance.
    go_ = to.grad.cpu().numpy().astype(np.float64)
    scale = float(np.abs(ref['grad_offset']).max())
    assert float((np.abs(go_ - ref['grad_offset']) > 5e-5 * scale).mean()) <= 1e-5


@pytest.mark.gpu
def test_cached_inference_pack_serves_every_map_size():
    """Inference caches the packed weight images per weight tensor (kgdet_amd/dcn.py pack_weight); the serial head
    applies ONE DeformConv weight to five pyramid levels.  A pack made for a map too large for the LDS-plane kernels
    must still carry the bf16 plane images the small levels read (it once did not: freshly allocated memory reads as
    zeros, recycled memory as whatever was there -- so the allocator's free blocks are filled with NaN first)."""
    _require_gpu()
    from kgdet_amd import dcn
    torch.manual_seed(3)
    w = torch.randn(256, 256, 3, 3, device='cuda') * 0.01
    maps = [(100, 168), (25, 42), (7, 11)]
    xs = [torch.randn(2, 256, h, ww, device='cuda') for h, ww in maps]
    offs = [torch.randn(2, 18, h, ww, device='cuda') for h, ww in maps]
    with torch.no_grad():
        want = []
        for x, o in zip(xs, offs):          # one pack per call: the reference
            dcn.clear_pack_cache()
            want.append(dcn.deform_conv_cat(x, [o], [w], [1]).clone())
        dcn.clear_pack_cache()
        junk = [torch.full((32 * 1024 * 1024,), float('nan'), device='cuda') for _ in range(8)]
        del junk
        got = [dcn.deform_conv_cat(x, [o], [w], [1]) for x, o in zip(xs, offs)]     # large map first, pack cached
    for g, wnt in zip(got, want):
        assert torch.isfinite(g).all()
        assert torch.equal(g, wnt)


@pytest.mark.gpu
def test_multi_weight_pack_equals_single_packs():
    """kgdet_dcn_pack_weight_multi (the six weights of a head stage in one launch) writes the same images as one
    kgdet_dcn_pack_weight per weight -- mixed kernel sizes, channel counts that need padding, a grouped weight and a
    7x7 (the two that take the one-call-each route inside the multi call)."""
    _require_gpu()
    from kgdet_amd import dcn
    torch.manual_seed(11)
    specs = [(256, 256, 3, 1, 1), (64, 256, 1, 1, 1), (40, 24, 3, 1, 1), (256, 256, 5, 1, 1), (32, 64, 3, 2, 1),
             (16, 32, 7, 1, 1), (256, 64, 3, 1, 4), (100, 36, 1, 1, 1), (256, 256, 3, 1, 1), (8, 8, 3, 1, 1)]
    ws, shapes = [], []
    for O, C, k, g, dg in specs:
        w = torch.randn(O, C // g, k, k, device='cuda')
        x = torch.empty(1, C, 9, 9, device='cuda')
        ws.append(w)
        shapes.append(dcn._shape(x, w, (1, 1), (k // 2, k // 2), (1, 1), g, dg))
    junk = [torch.full((8 * 1024 * 1024,), float('nan'), device='cuda') for _ in range(4)]
    del junk
    single = [dcn.pack_weight(w, s).clone() for w, s in zip(ws, shapes)]
    junk = [torch.full((8 * 1024 * 1024,), float('nan'), device='cuda') for _ in range(4)]
    del junk
    multi = dcn.pack_weights(ws, shapes)
    for a, b, spec in zip(single, multi, specs):
        assert a.shape == b.shape
        # compare the bytes (the images hold bf16 pairs viewed as fp32: NaN patterns are legal payloads); rows the
        # pack kernels never write are poisoned on both sides, so compare only where the single pack wrote
        ai, bi = a.view(torch.int32), b.view(torch.int32)
        assert torch.equal(ai, bi), spec


@pytest.mark.gpu
def test_random_shapes_split_kernels_agree_with_exact_fp32():
    """24 random problems (1-8 images, 16-256 channels, 16-512 filters, 1x1 ... 7x7 and non-square kernels, maps up to
    32 x 40, v1 and v2, offset scales 0.3-3 pixels): forward and every gradient of the split-operand plane kernels against
    the exact-fp32 kernels (`dcn.arithmetic('exact')`) to 1e-4 of each result's scale (measured: <= 9e-6).  Covers the
    schedules the fixed cases do not: stream-K slices that start inside a channel chunk, static ranges of uneven parts,
    first groups of one to four stages, tiles with dead pixel columns."""
    _require_gpu()
    from kgdet_amd import dcn
    rng = np.random.default_rng(1)
    kernels = [(1, 1), (3, 3), (3, 3), (5, 5), (7, 7), (3, 1), (1, 3)]
    for case in range(24):
        N = int(rng.integers(1, 9))
        C = int(rng.choice([16, 32, 48, 64, 96, 128, 256]))
        O = int(rng.choice([16, 32, 64, 128, 256, 512]))
        kh, kw = kernels[int(rng.integers(0, 7))]
        H, W = int(rng.integers(4, 33)), int(rng.integers(4, 41))
        v2 = bool(rng.integers(0, 2))
        sigma = float(rng.choice([0.3, 1.0, 3.0]))
        g = torch.Generator(device='cpu').manual_seed(case)
        x = torch.randn(N, C, H, W, generator=g).cuda()
        off = (torch.randn(N, 2 * kh * kw, H, W, generator=g) * sigma).cuda()
        w = (torch.randn(O, C, kh, kw, generator=g) * 0.05).cuda()
        m = torch.rand(N, kh * kw, H, W, generator=g).cuda() if v2 else None
        go = torch.randn(N, O, H, W, generator=g).cuda()
        res = {}
        for mode in ('split', 'exact'):
            xs, os_, ws = x.clone().requires_grad_(), off.clone().requires_grad_(), w.clone().requires_grad_()
            ms = m.clone().requires_grad_() if v2 else None
            with dcn.arithmetic(mode):
                if v2:
                    out = dcn.modulated_deform_conv(xs, os_, ms, ws, None, 1, (kh // 2, kw // 2), 1, 1, 1)
                else:
                    out = dcn.deform_conv(xs, os_, ws, 1, (kh // 2, kw // 2), 1, 1, 1)
                out.backward(go)
            res[mode] = [out.detach(), xs.grad, os_.grad, ws.grad] + ([ms.grad] if v2 else [])
        for name, a, b in zip(('out', 'grad_input', 'grad_offset', 'grad_weight', 'grad_mask'), res['split'], res['exact']):
            assert torch.isfinite(a).all(), (case, name)
            err = float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))
            assert err < 1e-4, (case, name, err, (N, C, O, kh, kw, H, W, v2, sigma))
