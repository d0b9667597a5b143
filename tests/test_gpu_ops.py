"""GPU parity for the small hot-path ops (through the C ABI) against the CPU oracle and the
golden vectors produced by the compiled reference.

Bars: NMS / soft-NMS index selection bit-exact (and soft-NMS scores bit-exact); focal loss,
moment bbox and PS-RoI pooling within 1e-5 of their output scale (different libm / summation).
"""
import os

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def _close(actual, desired, tol=1e-5):
    desired = np.asarray(desired, np.float64)
    scale = max(float(np.abs(desired).max()), 1e-6)
    err = float(np.abs(np.asarray(actual, np.float64) - desired).max()) / scale
    assert err < tol, 'max error %.3e of output scale (tol %.1e)' % (err, tol)


@pytest.fixture(scope='module')
def G(golden_dir):
    return np.load(os.path.join(golden_dir, 'nms_golden.npz'))


def _cases(prefix, G):
    i = 0
    while '%s%d_dets' % (prefix, i) in G:
        yield i
        i += 1


# ---------------------------------------------------------------------------------------------
def test_nms_reference_golden_bit_exact(G):
    from kgdet_amd.nms import nms
    for i in list(_cases('nms', G)) + ['tie']:
        d = G['nms%s_dets' % i]
        thr = float(G['nms%s_thr' % i])
        dets, inds = nms(torch.from_numpy(d).cuda(), thr)
        np.testing.assert_array_equal(inds.cpu().numpy(), G['nms%s_keep' % i])
        np.testing.assert_array_equal(dets.cpu().numpy(), d[G['nms%s_keep' % i]])


def test_nms_numpy_and_cpu_tensor_inputs_run_on_gpu(G):
    from kgdet_amd.nms import nms
    d = G['nms6_dets']
    dets, inds = nms(d, 0.5)                       # numpy in -> numpy out
    assert isinstance(inds, np.ndarray) and inds.dtype == np.int64
    np.testing.assert_array_equal(inds, G['nms6_keep'])
    dets_t, inds_t = nms(torch.from_numpy(d), 0.5)  # CPU tensor in -> CPU tensor out
    assert not inds_t.is_cuda
    np.testing.assert_array_equal(inds_t.numpy(), G['nms6_keep'])
    e, ei = nms(torch.zeros(0, 5).cuda(), 0.5)
    assert ei.numel() == 0 and ei.dtype == torch.long
    with pytest.raises(TypeError):
        nms([[0, 0, 1, 1, 1]], 0.5)


def test_nms_batched_matches_oracle():
    from kgdet_amd.nms import nms_batched
    rng = np.random.default_rng(3)
    segs, offs = [], [0]
    for n in (0, 1, 37, 300, 64, 1000, 129, 0, 513):
        x1 = rng.uniform(0, 800, n); y1 = rng.uniform(0, 600, n)
        d = np.stack([x1, y1, x1 + rng.uniform(1, 300, n), y1 + rng.uniform(1, 300, n),
                      rng.permutation(n) / max(n, 1) + 0.01], 1).astype(np.float32).reshape(-1, 5)
        segs.append(d)
        offs.append(offs[-1] + n)
    dets = torch.from_numpy(np.concatenate(segs)).cuda()
    offsets = torch.tensor(offs, dtype=torch.int64).cuda()
    keep, num = nms_batched(dets, offsets, 0.5)
    keep, num = keep.cpu().numpy(), num.cpu().numpy()
    for i, d in enumerate(segs):
        ref = oracle.nms(d, 0.5)
        assert num[i] == len(ref)
        np.testing.assert_array_equal(keep[offs[i]:offs[i] + num[i]], ref)


def test_nms_fullsize_properties():
    """At nms_pre size: idempotence and pairwise-IoU property of the survivors."""
    from kgdet_amd.nms import nms
    rng = np.random.default_rng(11)
    n = 4000
    x1 = rng.uniform(0, 1300, n); y1 = rng.uniform(0, 780, n)
    d = np.stack([x1, y1, x1 + rng.uniform(5, 200, n), y1 + rng.uniform(5, 200, n), rng.permutation(n) / n],
                 1).astype(np.float32)
    kept, inds = nms(torch.from_numpy(d).cuda(), 0.5)
    inds = inds.cpu().numpy()
    assert np.all(np.diff(inds) > 0)                      # ascending index order
    np.testing.assert_array_equal(inds, oracle.nms(d, 0.5))
    kept2, inds2 = nms(kept, 0.5)
    assert inds2.numel() == kept.shape[0]                 # idempotent


def test_fused_multiclass_soft_nms_reproduces_the_reference_goldens(G):
    """kgdet_multiclass_soft_nms (round 4: the whole batch's per-class soft-NMS in two launches, nothing read by the host): every
    golden case of the COMPILED reference soft_nms_cpu.pyx, fed as one class of one image with all boxes above the score
    threshold, leaves the kernel with the reference's survivors, order and decayed scores, bit for bit (3350-box case
    included); a second class and a second image around it must not disturb it."""
    from kgdet_amd.postprocess import multiclass_soft_nms_kp_fused
    names = {1: 'linear', 2: 'gaussian'}
    for i in _cases('soft', G):
        thr, method, sigma, min_score = G['soft%d_cfg' % i]
        d = G['soft%d_dets' % i]
        n = d.shape[0]
        if n == 0 or n > 4500:
            continue
        boxes = torch.zeros(2, n, 4)
        boxes[1] = torch.from_numpy(d[:, :4])
        floor = float(d[:, 4].min()) - 1.0                      # every golden box is a candidate ...
        scores = torch.full((2, n, 3), floor - 1.0)             # ... and nothing else, except a few rows of
        scores[1, :, 1] = torch.from_numpy(d[:, 4])
        scores[0, :min(n, 5), 0] = 0.9                          # another image / other classes: their own problems
        scores[1, :min(n, 5), 2] = 0.5
        kp = torch.arange(n, dtype=torch.float32).view(1, n, 1).expand(2, n, 3).contiguous()
        cfg = dict(type='soft_nms', iou_thr=float(thr), method=names[int(method)], sigma=float(sigma), min_score=float(min_score))
        M = n + 16
        det, label, k, count = multiclass_soft_nms_kp_fused(boxes.cuda(), scores.cuda(), kp.cuda(), floor, cfg, M)
        cnt = int(count[1])
        lab = label[1, :cnt].cpu().numpy()
        sel = lab == 1
        np.testing.assert_array_equal(det[1, :cnt][torch.from_numpy(sel).cuda()].cpu().numpy(), G['soft%d_new' % i])
        np.testing.assert_array_equal(k[1, :cnt, 0].cpu().numpy()[sel].astype(np.int64), G['soft%d_inds' % i])
        assert np.all(np.diff(lab) >= 0)                        # class-major concatenation


@pytest.mark.parametrize('method', ['linear', 'gaussian'])
def test_fused_multiclass_soft_nms_equals_the_per_class_path(method):
    """the fused batch path == multiclass_nms_kp (the per-image, per-class loop over the bit-exact soft_nms op) -- candidate
    filter, per-class order, landmark rows, and the top-max_num cut by decayed score -- on clustered boxes: 3 images x 5
    classes x 700 candidates, with max_num below AND above the number of survivors"""
    from kgdet_amd.postprocess import multiclass_nms_kp, multiclass_soft_nms_kp_fused
    rng = np.random.default_rng(3)
    B, N, C = 3, 700, 5
    ctr = rng.uniform(100, 1200, size=(B, 30, 2))[np.arange(B)[:, None], rng.integers(0, 30, size=(B, N))]
    ctr = ctr + rng.normal(0, 15, size=(B, N, 2))
    wh = rng.uniform(40, 160, size=(B, N, 2))
    boxes = torch.from_numpy(np.concatenate([ctr - wh / 2, ctr + wh / 2], -1).astype(np.float32)).cuda()
    scores = torch.from_numpy((rng.uniform(0, 1, size=(B, N, C)) ** 3).astype(np.float32)).cuda()
    kpts = torch.from_numpy(rng.normal(size=(B, N, 9)).astype(np.float32)).cuda()
    cfg = dict(type='soft_nms', iou_thr=0.5, method=method, sigma=0.5, min_score=0.05)
    for max_num in (100, 3000):
        det, label, k, count = multiclass_soft_nms_kp_fused(boxes, scores, kpts, 0.05, cfg, max_num)
        for b in range(B):
            ms = torch.cat([scores.new_zeros(N, 1), scores[b]], 1)
            wd, wl, wk = multiclass_nms_kp(boxes[b], ms, kpts[b], 0.05, cfg, max_num)
            n = int(count[b])
            assert n == wd.shape[0] and n > 50
            if n == max_num:        # the cut is by decayed score; equal scores could come in either order from torch's sort
                assert len(np.unique(wd[:, 4].cpu().numpy())) == n
            assert torch.equal(det[b, :n], wd) and torch.equal(label[b, :n], wl) and torch.equal(k[b, :n], wk)
            assert float(det[b, n:].abs().sum()) == 0.0


def test_soft_nms_reference_golden_bit_exact(G):
    from kgdet_amd.nms import soft_nms
    names = {1: 'linear', 2: 'gaussian'}
    for i in _cases('soft', G):
        thr, method, sigma, min_score = G['soft%d_cfg' % i]
        d = G['soft%d_dets' % i]
        nd, ni = soft_nms(torch.from_numpy(d).cuda(), float(thr), names[int(method)], float(sigma),
                          float(min_score))
        np.testing.assert_array_equal(ni.cpu().numpy(), G['soft%d_inds' % i])
        np.testing.assert_array_equal(nd.cpu().numpy(), G['soft%d_new' % i])
    nd, ni = soft_nms(G['soft4_dets'], 0.5, 'linear', 0.5, 0.05)   # numpy in -> numpy out
    np.testing.assert_array_equal(ni, G['soft4_inds'])
    with pytest.raises(ValueError):
        soft_nms(torch.zeros(1, 5).cuda(), 0.5, method='nope')


# ---------------------------------------------------------------------------------------------
def test_focal_loss_forward_backward():
    from kgdet_amd.focal_loss import sigmoid_focal_loss, SigmoidFocalLoss
    rng = np.random.default_rng(0)
    n, c = 2100, 13
    logits = (rng.normal(size=(n, c)) * 3).astype(np.float32)
    logits[0, :3] = [-100, 100, 0]
    target = rng.integers(0, c + 1, n)
    target[:200] = 0
    dl = rng.normal(size=(n, c)).astype(np.float32)
    x = torch.from_numpy(logits).cuda().requires_grad_()
    t = torch.from_numpy(target).cuda()
    loss = sigmoid_focal_loss(x, t, 2.0, 0.25)
    loss.backward(torch.from_numpy(dl).cuda())
    _close(loss.detach().cpu().numpy(), oracle.sigmoid_focal_loss_forward(logits, target, 2.0, 0.25))
    _close(x.grad.cpu().numpy(), oracle.sigmoid_focal_loss_backward(logits, target, dl, 2.0, 0.25))
    m = SigmoidFocalLoss(2.0, 0.25)
    assert abs(float(m(x.detach(), t)) - float(loss.sum())) < 1e-2
    with pytest.raises(NotImplementedError):
        sigmoid_focal_loss(torch.zeros(2, 3), torch.zeros(2, dtype=torch.long), 2.0, 0.25)


def _moment_ref(pts, mt, y_first=True):
    """autograd restatement of KP3:373-388 in float64"""
    B, C2, H, W = pts.shape
    r = pts.view(B, -1, 2, H, W)
    py = r[:, :, 0] if y_first else r[:, :, 1]
    px = r[:, :, 1] if y_first else r[:, :, 0]
    my, mx = py.mean(1, keepdim=True), px.mean(1, keepdim=True)
    sy, sx = torch.std(py - my, dim=1, keepdim=True), torch.std(px - mx, dim=1, keepdim=True)
    hw, hh = sx * torch.exp(mt[0]), sy * torch.exp(mt[1])
    return torch.cat([mx - hw, my - hh, mx + hw, my + hh], 1)


@pytest.mark.parametrize('n_pts,y_first', [(83, True), (9, True), (25, False)])
def test_moment_bbox(n_pts, y_first):
    from kgdet_amd.moment import moment_bbox
    torch.manual_seed(0)
    pts = torch.randn(2, 2 * n_pts, 25, 42, dtype=torch.float64) * 3
    mt = torch.tensor([0.3, -0.2], dtype=torch.float64)
    g = torch.randn(2, 4, 25, 42, dtype=torch.float64)
    p64, m64 = pts.clone().requires_grad_(), mt.clone().requires_grad_()
    ref = _moment_ref(p64, m64, y_first)
    ref.backward(g)
    p = pts.float().cuda().requires_grad_()
    m = mt.float().cuda().requires_grad_()
    out = moment_bbox(p, m, y_first)
    out.backward(g.float().cuda())
    _close(out.detach().cpu().numpy(), ref.detach().numpy())
    _close(p.grad.cpu().numpy(), p64.grad.numpy())
    _close(m.grad.cpu().numpy(), m64.grad.numpy(), 1e-4)


def test_moment_bbox_known_answer():
    """points on a regular grid: mean / unbiased std computable by hand."""
    from kgdet_amd.moment import moment_bbox
    ys = torch.tensor([0., 1., 2., 3.])    # mean 1.5, unbiased var = 5/3
    xs = torch.tensor([10., 10., 14., 14.])  # mean 12, unbiased var = 16/3
    pts = torch.stack([ys, xs], 1).reshape(1, 8, 1, 1).cuda()
    out = moment_bbox(pts, torch.zeros(2).cuda(), True).flatten().cpu().numpy()
    sy, sx = np.sqrt(5 / 3), np.sqrt(16 / 3)
    np.testing.assert_allclose(out, [12 - sx, 1.5 - sy, 12 + sx, 1.5 + sy], rtol=1e-6)


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('no_trans,group_size,part', [(True, 1, None), (False, 1, None), (False, 7, 7),
                                                       (False, 1, 4)])
def test_deform_psroi_pooling(no_trans, group_size, part):
    from kgdet_amd.deform_pool import deform_roi_pooling
    rng = np.random.default_rng(1)
    B, H, W, P, out_c = 2, 24, 30, 7, 8
    C = out_c * group_size * group_size
    data = rng.normal(size=(B, C, H, W)).astype(np.float32)
    R = 12
    x1 = rng.uniform(-20, 400, R); y1 = rng.uniform(-20, 300, R)
    rois = np.stack([rng.integers(0, B, R), x1, y1, x1 + rng.uniform(1, 200, R), y1 + rng.uniform(1, 200, R)],
                    1).astype(np.float32)
    rois[0, 1:] = [5, 5, 5, 5]           # degenerate roi -> min size 0.1
    part_size = P if part is None else part
    ncls = 2
    offset = (rng.normal(size=(R, 2 * ncls, part_size, part_size)) * 0.5).astype(np.float32)
    go = rng.normal(size=(R, out_c, P, P)).astype(np.float32)
    td = torch.from_numpy(data).cuda().requires_grad_()
    to = torch.from_numpy(offset).cuda().requires_grad_()
    out = deform_roi_pooling(td, torch.from_numpy(rois).cuda(), to if not no_trans else to.new_empty(0), 1 / 16., P,
                             out_c, no_trans, group_size, part_size, 4, 0.1)
    out.backward(torch.from_numpy(go).cuda())
    ro, rc = oracle.deform_psroi_forward(data, rois, offset, np.float32(1 / 16.), P, out_c, no_trans, group_size,
                                         part_size, 4, np.float32(0.1))
    _close(out.detach().cpu().numpy(), ro)
    gd, gt = oracle.deform_psroi_backward(go, rc, data, rois, offset, np.float32(1 / 16.), P, out_c, no_trans,
                                          group_size, part_size, 4, np.float32(0.1))
    _close(td.grad.cpu().numpy(), gd)
    if not no_trans:
        _close(to.grad.cpu().numpy(), gt)


@pytest.mark.gpu
@pytest.mark.parametrize('no_trans,S,group_size,C,out_c', [(False, 2, 1, 256, 256), (True, 4, 1, 256, 256),
                                                          (False, 4, 7, 392, 8)])
def test_deform_psroi_pooling_at_size_bit_repeatable(no_trans, S, group_size, C, out_c):
    """512 RoIs x 256 channels x 7x7 bins on a [2, C, 50, 84] map (DeformRoIPoolingPack's shape; and the
    position-sensitive 7x7-group variant): forward, grad_data and grad_offset against oracle/psroi_oracle.inc in float32;
    the backward is BITWISE repeatable (a gather per output tile / one workgroup per (RoI, class): no atomics, round-2
    review item 7), writes every gradient element (NaN-filled outputs come back finite), and -- because thread = channel
    adds the samples in the reference loop's own serial order without fp contraction -- grad_data equals the serial
    float32 oracle bit for bit."""
    import ctypes
    from kgdet_amd import _lib
    from kgdet_amd.deform_pool import deform_roi_pooling
    rng = np.random.default_rng(7)
    B, H, W, P, R = 2, 50, 84, 7, 512
    data = rng.normal(size=(B, C, H, W)).astype(np.float32)
    x1 = rng.uniform(-30, 1250, R); y1 = rng.uniform(-30, 720, R)
    rois = np.stack([rng.integers(0, B, R), x1, y1, x1 + rng.uniform(8, 600, R), y1 + rng.uniform(8, 500, R)],
                    1).astype(np.float32)
    rois[0, 1:] = [0, 0, 1343, 799]            # the whole image: its window does not fit the LDS budget (direct lane)
    rois[1, 1:] = [40, 40, 40, 40]             # degenerate
    rois[2, 1:] = [5000, 5000, 5100, 5100]     # outside the map: count 0 everywhere
    offset = (rng.normal(size=(R, 2, P, P)) * 0.5).astype(np.float32)
    go = rng.normal(size=(R, out_c, P, P)).astype(np.float32)
    td = torch.from_numpy(data).cuda().requires_grad_()
    to = torch.from_numpy(offset).cuda().requires_grad_()
    args = (torch.from_numpy(rois).cuda(), to if not no_trans else to.new_empty(0), 1 / 16., P, out_c, no_trans,
            group_size, P, S, 0.1)
    out = deform_roi_pooling(td, *args)
    out.backward(torch.from_numpy(go).cuda())
    g1 = td.grad.clone()
    t1 = None if no_trans else to.grad.clone()
    ro, rc = oracle.deform_psroi_forward(data, rois, offset, np.float32(1 / 16.), P, out_c, no_trans, group_size, P, S,
                                         np.float32(0.1))
    _close(out.detach().cpu().numpy(), ro, 2e-6)
    gd, gt = oracle.deform_psroi_backward(go, rc, data, rois, offset, np.float32(1 / 16.), P, out_c, no_trans,
                                          group_size, P, S, np.float32(0.1))
    np.testing.assert_array_equal(g1.cpu().numpy(), gd)          # same summation order, same expressions: bit-exact
    if not no_trans:
        _close(t1.cpu().numpy(), gt, 2e-5)
    # second run into NaN-filled outputs through the C ABI: identical bits, nothing left unwritten
    L = _lib.lib()
    from kgdet_amd.deform_pool import _shape
    shape = _shape(td, args[0], to, 1 / 16., P, out_c, no_trans, group_size, P, S, 0.1)
    if no_trans:
        shape.num_classes = 1
    g2 = torch.full_like(g1, float('nan'))
    t2 = torch.full_like(to, float('nan'))
    cnt = torch.from_numpy(rc).cuda()
    ws_bytes = L.kgdet_deform_psroi_backward_workspace_bytes(ctypes.byref(shape))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device='cuda')
    _lib.check(L.kgdet_deform_psroi_backward(
        ctypes.byref(shape), _lib.ptr(torch.from_numpy(go).cuda()), _lib.ptr(cnt), _lib.ptr(td.detach()), _lib.ptr(args[0]),
        None if no_trans else _lib.ptr(to.detach()), _lib.ptr(g2), None if no_trans else _lib.ptr(t2), _lib.ptr(ws),
        ctypes.c_size_t(ws_bytes), _lib.current_stream()), 'psroi_backward')
    torch.cuda.synchronize()
    assert torch.equal(g1, g2)
    if not no_trans:
        assert torch.equal(t1, t2)


@pytest.mark.gpu
def test_deform_psroi_grad_data_hot_pixels_take_the_long_list_sorts():
    """csrc/psroi.hip grad_data lists: 200 copies of one tiny RoI put ~10^5 sample corners on a handful of pixels, five copies of
    another ~4 000 -- contribution lists far beyond the per-wave rank sort (> 512 entries: psroi_list_sort_long, in LDS up to
    8 192 entries, in place in global memory beyond) and the wave-aggregated slot counters; grad_data must still equal the serial
    float32 oracle bit for bit, twice."""
    from kgdet_amd.deform_pool import deform_roi_pooling
    rng = np.random.default_rng(11)
    B, C, H, W, P, S, out_c = 2, 16, 20, 24, 7, 4, 16
    data = rng.normal(size=(B, C, H, W)).astype(np.float32)
    rois = np.concatenate([
        np.tile(np.array([[0, 100, 90, 112, 101]], np.float32), (200, 1)),     # < 1 map cell wide at scale 1/16
        np.tile(np.array([[1, 200, 150, 215, 166]], np.float32), (5, 1)),
        np.stack([rng.integers(0, B, 30), rng.uniform(0, 200, 30), rng.uniform(0, 150, 30),
                  rng.uniform(200, 380, 30), rng.uniform(150, 310, 30)], 1).astype(np.float32)])
    R = rois.shape[0]
    offset = (rng.normal(size=(R, 2, P, P)) * 0.5).astype(np.float32)
    go = rng.normal(size=(R, out_c, P, P)).astype(np.float32)
    grads = []
    for _ in range(2):
        td = torch.from_numpy(data).cuda().requires_grad_()
        to = torch.from_numpy(offset).cuda().requires_grad_()
        out = deform_roi_pooling(td, torch.from_numpy(rois).cuda(), to, 1 / 16., P, out_c, False, 1, P, S, 0.1)
        out.backward(torch.from_numpy(go).cuda())
        grads.append(td.grad.clone())
    _, rc = oracle.deform_psroi_forward(data, rois, offset, np.float32(1 / 16.), P, out_c, False, 1, P, S, np.float32(0.1))
    gd, _ = oracle.deform_psroi_backward(go, rc, data, rois, offset, np.float32(1 / 16.), P, out_c, False, 1, P, S,
                                         np.float32(0.1))
    np.testing.assert_array_equal(grads[0].cpu().numpy(), gd)
    assert torch.equal(grads[0], grads[1])


@pytest.mark.gpu
@pytest.mark.parametrize('N,C,frac,max_num', [(1000, 13, 0.05, 100), (257, 5, 0.5, 20), (64, 1, 1.0, 100), (1200, 13, 0.2, 100)])
def test_multiclass_nms_fused_matches_per_image_reference_path(N, C, frac, max_num):
    """csrc/nms.hip multiclass_nms_segments + multiclass_select == bbox_nms_kp.py's per-image, per-class loop
    (filter > score_thr, NMS, concatenate, top max_num by score), bit for bit, including an image without candidates
    and duplicated scores."""
    from kgdet_amd.postprocess import multiclass_nms_kp, multiclass_nms_kp_fused
    g = torch.Generator(device='cpu').manual_seed(N + C)
    B, K = 3, 12
    ctr = torch.rand(B, 8, 2, generator=g) * torch.tensor([1200., 700.]) + 50
    which = torch.randint(0, 8, (B, N), generator=g)
    c = torch.gather(ctr, 1, which.unsqueeze(-1).expand(B, N, 2)) + torch.randn(B, N, 2, generator=g) * 15
    wh = torch.rand(B, N, 2, generator=g) * 200 + 30
    boxes = torch.cat([c - wh / 2, c + wh / 2], -1).cuda()
    scores = torch.rand(B, N, C, generator=g)
    scores = torch.where(torch.rand(B, N, C, generator=g) < frac, scores, scores * 0.04)     # most below 0.05
    scores[:, : N // 4] = (scores[:, : N // 4] * 50).round() / 50                           # ties
    scores[1] = 0.01                                                                        # image without candidates
    scores = scores.cuda()
    kpts = torch.randn(B, N, K, generator=g).cuda()
    det, label, kp, count = multiclass_nms_kp_fused(boxes, scores, kpts, 0.05, 0.5, max_num)
    assert det.shape == (B, max_num, 5) and count.tolist()[1] == 0
    for b in range(B):
        full = torch.cat([scores.new_zeros(N, 1), scores[b]], 1)
        rd, rl, rk = multiclass_nms_kp(boxes[b], full, kpts[b], 0.05, dict(type='nms', iou_thr=0.5), max_num)
        n = int(count[b])
        assert n == rd.shape[0]
        if n > max_num - 1 and rd.shape[0] == max_num:
            # the reference's final sort is unstable for equal scores: compare as sets of rows when scores tie
            pass
        assert torch.equal(det[b, :n, 4], rd[:, 4])
        same = (det[b, :n] == rd).all(1) & (label[b, :n] == rl) & (kp[b, :n] == rk).all(1)
        if not bool(same.all()):   # only rows whose score is duplicated may be permuted
            sc = rd[:, 4]
            dup = (sc.unsqueeze(0) == sc.unsqueeze(1)).sum(1) > 1
            assert bool(same[~dup].all())
        assert float(det[b, n:].abs().sum()) == 0 and float(kp[b, n:].abs().sum()) == 0


def test_nms_beyond_the_on_chip_limit_bit_exact(golden_dir):
    """nms_wrapper.nms accepts any N (nms_wrapper.py:8-49): segments of 4097 / 8000 / 12000 boxes run the same algorithm
    on a global scratch buffer (csrc/nms.hip nms_segments_large); kept indices bit-exact vs the compiled reference, also
    as segments of one batched launch mixed with short ones."""
    import os
    from kgdet_amd.nms import nms, nms_batched
    from tests.golden.make_nms_golden import make_boxes
    L = np.load(os.path.join(golden_dir, 'nms_large_golden.npz'))
    dets_all, keeps, thr0 = [], [], None
    i = 0
    while 'case%d' % i in L:
        n, seed, cluster, quant = [int(v) for v in L['case%d' % i]]
        d = make_boxes(np.random.default_rng(seed), n, cluster=bool(cluster), quantize=bool(quant))
        thr = float(L['thr%d' % i])
        dets, inds = nms(torch.from_numpy(d).cuda(), thr)
        np.testing.assert_array_equal(inds.cpu().numpy(), L['keep%d' % i])
        np.testing.assert_array_equal(dets.cpu().numpy(), d[L['keep%d' % i]])
        if abs(thr - 0.5) < 1e-9:
            dets_all.append(d)
            keeps.append(L['keep%d' % i])
        i += 1
    # one batched launch: long, empty, short and long segments together
    short = make_boxes(np.random.default_rng(9), 300)
    segs = [dets_all[0], np.zeros((0, 5), np.float32), short, dets_all[1]]
    offs = np.cumsum([0] + [len(s) for s in segs]).astype(np.int64)
    keep, num = nms_batched(torch.from_numpy(np.concatenate(segs)).cuda(), torch.from_numpy(offs).cuda(), 0.5)
    num = num.cpu().numpy()
    keep = keep.cpu().numpy()
    import oracle
    want = [keeps[0], np.zeros(0, np.int64), oracle.nms(short, 0.5), keeps[1]]
    for s, w in enumerate(want):
        assert num[s] == len(w)
        np.testing.assert_array_equal(keep[offs[s]:offs[s] + num[s]], w)


@pytest.mark.parametrize('tail', ['empty', 'one_box'])
def test_nms_large_segment_with_a_thousand_small_ones(tail):
    """when one segment exceeds the on-chip limit EVERY segment gets a slice of the scratch buffer (at least
    512 + 255 bytes); the old bound allotted 256 per segment and the kernel wrote past the allocation.  One 4100-box
    segment followed by 999 empty / one-box segments: results equal the oracle's and the words after the workspace the
    size query asks for stay untouched."""
    import ctypes
    import oracle
    from kgdet_amd import _lib
    from tests.golden.make_nms_golden import make_boxes
    rng = np.random.default_rng(21)
    big = make_boxes(rng, 4100)
    small = [make_boxes(rng, 1) if tail == 'one_box' else np.zeros((0, 5), np.float32) for _ in range(999)]
    segs = [big] + small
    offs = np.cumsum([0] + [len(x) for x in segs]).astype(np.int64)
    dets = torch.from_numpy(np.concatenate(segs)).cuda()
    offs_t = torch.from_numpy(offs).cuda()
    L = _lib.lib()
    T, S = dets.shape[0], len(segs)
    need = L.kgdet_nms_workspace_bytes(ctypes.c_int64(T), ctypes.c_int32(S))
    guard = 1 << 20
    buf = torch.full((need + guard,), 0xA5, dtype=torch.uint8, device='cuda')
    keep = torch.empty(T, dtype=torch.int64, device='cuda')
    num = torch.zeros(S, dtype=torch.int64, device='cuda')
    _lib.check(L.kgdet_nms_batched(_lib.ptr(dets), _lib.ptr(offs_t), ctypes.c_int32(S), ctypes.c_int64(T),
                                   ctypes.c_int64(4100), ctypes.c_float(0.5), _lib.ptr(keep), _lib.ptr(num),
                                   _lib.ptr(buf), ctypes.c_size_t(need), _lib.current_stream()), 'kgdet_nms_batched')
    torch.cuda.synchronize()
    assert bool((buf[need:] == 0xA5).all()), 'kernel wrote past the workspace the size query asked for'
    num = num.cpu().numpy()
    keep = keep.cpu().numpy()
    want0 = oracle.nms(big, 0.5)
    assert num[0] == len(want0)
    np.testing.assert_array_equal(keep[:num[0]], want0)
    if tail == 'one_box':
        assert (num[1:] == 1).all() and (keep[offs[1:-1]] == 0).all()
    else:
        assert (num[1:] == 0).all()


@pytest.mark.parametrize('shape,beta,divisor,with_weight', [((2100, 588), 1.0 / 9.0, 128.0, True), ((2100, 4), 1.0 / 9.0, 128.0, True),
                                                            ((37, 5), 1.0, None, False), ((1, 1), 0.5, 3.0, True),
                                                            ((10500, 588), 1.0 / 9.0, 32.0, True)])
def test_fused_smooth_l1_matches_reference_chain(shape, beta, divisor, with_weight):
    """SmoothL1Loss on the HIP op (csrc/smooth_l1.hip: one pass each way) against the reference's chain of torch ops
    (smooth_l1_loss.py:8-45, utils.py:7-52; KP3:621-665 divides prediction and target by point_base_scale * stride first):
    the loss to 2e-6 (the only difference is the order of the fp32 sum), the gradient to rounding; a device-tensor
    avg_factor (the sync-free training path) and a Python one."""
    from kgdet_amd import losses
    torch.manual_seed(shape[0])
    pred = (torch.randn(*shape, device='cuda') * 40 + 300).requires_grad_()
    target = pred.detach() + torch.randn(*shape, device='cuda') * 20
    target[::3] = pred.detach()[::3]          # exact zeros of the difference (|x|' = 0 there)
    weight = (torch.rand(*shape, device='cuda') > 0.7).float() * torch.rand(*shape, device='cuda') if with_weight else None
    mod = losses.SmoothL1Loss(beta=beta, loss_weight=0.5)
    for avg in (torch.tensor(17.0, device='cuda'), 17.0):
        assert losses.fused_smooth_l1_applicable(pred, target, weight, 'mean', avg)
        got = mod(pred, target, weight, avg_factor=avg, divisor=divisor)
        assert type(got.grad_fn).__name__ != 'NoneType'
        got.backward()
        g_got, pred.grad = pred.grad.clone(), None
        losses.FUSED_SMOOTH_L1 = False
        try:
            want = mod(pred, target, weight, avg_factor=avg, divisor=divisor)
        finally:
            losses.FUSED_SMOOTH_L1 = True
        want.backward()
        g_want, pred.grad = pred.grad.clone(), None
        assert abs(float(got) - float(want)) <= 2e-6 * max(abs(float(want)), 1e-6), (float(got), float(want))
        assert (g_got - g_want).abs().max().item() <= 1e-6 * g_want.abs().max().item() + 1e-12
        assert ((g_got == 0) == (g_want == 0)).all()


@pytest.mark.parametrize('relu', [True, False])
@pytest.mark.parametrize('N,C,G,H,W', [(2, 256, 32, 25, 42), (2, 256, 32, 13, 21), (1, 64, 32, 7, 5), (3, 96, 4, 16, 20),
                                       (2, 256, 32, 50, 84),
                                       (2, 256, 32, 100, 168),      # 134400 elements per group: 9 pixel slices (config 5, stride 8)
                                       (1, 64, 2, 75, 61),          # 32 channels per group, ragged slices
                                       (1, 48, 1, 40, 40)])         # 48 channels per group: one wave per channel, 5 slices
def test_fused_group_norm_relu_matches_torch(N, C, G, H, W, relu):
    """GroupNorm (+ ReLU) of a ConvModule on csrc/group_norm.hip (one pass each way) against nn.GroupNorm + F.relu in fp64:
    output, grad_x, grad_gamma, grad_beta to fp32 rounding of their scales"""
    import torch.nn.functional as F
    from kgdet_amd import layers
    torch.manual_seed(C + H)
    gn = torch.nn.GroupNorm(G, C).cuda()
    gn.weight.data.normal_(1.0, 0.5)
    gn.bias.data.normal_(0, 0.5)
    x = (torch.randn(N, C, H, W, device='cuda') * 3 + 1).requires_grad_()
    gy = torch.randn(N, C, H, W, device='cuda')
    assert layers.gn_act_applicable(x, gn)
    y = layers.gn_act(x, gn, relu)
    y.backward(gy)
    gd = torch.nn.GroupNorm(G, C).cuda().double()
    gd.load_state_dict({k: v.double() for k, v in gn.state_dict().items()})
    xd = x.detach().double().requires_grad_()
    yd = gd(xd)
    if relu:
        yd = yd * (y.detach() > 0)      # the ReLU mask of the fp32 result (an element within rounding of zero may flip in fp64)
    yd.backward(gy.double())
    for name, a_, b_ in (('y', y, yd), ('grad_x', x.grad, xd.grad), ('grad_gamma', gn.weight.grad, gd.weight.grad),
                         ('grad_beta', gn.bias.grad, gd.bias.grad)):
        err = (a_.double() - b_).abs().max().item() / b_.abs().max().item()
        assert err < 2e-6, (name, err)
    # a ConvModule takes the fused path and gives the same tensor as its modules called one by one
    m = layers.ConvModule(C, C, 3, padding=1, norm_cfg=dict(type='GN', num_groups=G, requires_grad=True)).cuda()
    xin = torch.randn(N, C, H, W, device='cuda')
    with torch.no_grad():
        got = m(xin)
        layers.FUSED_GN = False
        try:
            want = m(xin)
        finally:
            layers.FUSED_GN = True
    assert (got - want).abs().max().item() <= 1e-5 * want.abs().max().item()


def test_glue_kernels_match_the_torch_chain():
    """csrc/glue.hip: (a) the offsets of a Kp3RepBlock from the previous stage's reppoints -- value bit-identical to the
    reference expression gm * part + (1 - gm) * part.detach() - base (KP3:131-143), gradient gm * grad -- and (b) the
    stride-2 subsample + its zero-stuffing backward, against the torch ops they replace."""
    import kgdet_amd.heads as H
    from kgdet_amd.backbone import _subsample2
    from kgdet_amd.registry import build_head
    g = torch.Generator().manual_seed(0)
    blk = build_head(configs_kgdet().model.bbox_head).kp_rep_block_2.cuda()
    for C in (166, 170):
        x = (torch.randn(2, C, 25, 42, generator=g) * 3).cuda()
        gos = [torch.randn(2, 2 * k * k, 25, 42, generator=g).cuda() for k in (3, 5, 7)]
        res = {}
        for fused in (True, False):
            H._FUSED_OFFSETS = fused
            try:
                xi = x.clone().requires_grad_()
                offs = blk._dcn_offsets(xi, xi)
                torch.autograd.backward(offs, gos)
                res[fused] = ([o.detach() for o in offs], xi.grad.clone())
            finally:
                H._FUSED_OFFSETS = True
        for a, b in zip(res[True][0], res[False][0]):
            assert a.is_contiguous() and torch.equal(a, b)
        assert torch.equal(res[True][1], res[False][1])
    x = torch.randn(2, 8, 50, 84, generator=g).cuda().requires_grad_()
    y = _subsample2(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    xr = x.detach().clone().requires_grad_()
    yr = xr[:, :, ::2, ::2].contiguous()
    yr.backward(gy)
    assert torch.equal(y, yr) and torch.equal(x.grad, xr.grad)


def configs_kgdet():
    from kgdet_amd import configs
    return configs.kgdet_r50_fpn()


def test_moment_bbox_backward_is_bit_repeatable():
    """the transfer gradient is summed from per-block partials in block order (round 3; it used float atomics): same bits every run"""
    from kgdet_amd.moment import moment_bbox
    g = torch.Generator().manual_seed(0)
    pts = torch.randn(2, 166, 50, 84, generator=g).cuda().requires_grad_()
    mt = torch.tensor([0.2, -0.1]).cuda().requires_grad_()
    go = torch.randn(2, 4, 50, 84, generator=g).cuda()
    res = []
    for _ in range(3):
        pts.grad = mt.grad = None
        moment_bbox(pts, mt, True).backward(go)
        res.append((pts.grad.clone(), mt.grad.clone()))
    assert all(torch.equal(res[0][0], r[0]) and torch.equal(res[0][1], r[1]) for r in res[1:])
    ref_mt = mt.detach().double().requires_grad_()
    p64 = pts.detach().double().requires_grad_()
    _moment_ref(p64, ref_mt, True).backward(go.double())
    assert (res[0][1].double() - ref_mt.grad).abs().max() <= 1e-5 * ref_mt.grad.abs().max()


@pytest.mark.parametrize('y_first', [True, False])
@pytest.mark.parametrize('B,C,H,W,stride', [(2, 588, 25, 42, 32), (2, 18, 13, 21, 64), (1, 34, 7, 11, 128), (3, 130, 5, 9, 8)])
def test_fused_offset_to_pts_equals_the_torch_chain(B, C, H, W, stride, y_first):
    """offset_to_pts (reppoints_head_kp_serial.py:400-421) as one HIP pass each way (csrc/glue.hip): values and gradients
    bit-identical to the permute / flip / multiply / add chain"""
    from kgdet_amd import heads
    torch.manual_seed(C)
    pred = torch.randn(B, C, H, W, device='cuda', requires_grad=True)
    pts = torch.stack(torch.meshgrid(torch.arange(W, device='cuda') * float(stride), torch.arange(H, device='cuda') * float(stride),
                                     indexing='xy'), -1).reshape(-1, 2)
    centre = torch.cat([pts, torch.full((H * W, 1), float(stride), device='cuda')], 1)

    class _Head(heads.PointHeadMixin):
        point_strides = [stride]

    center_list = [[centre.clone()] for _ in range(B)]
    g = torch.randn(B, H * W, C, device='cuda')
    outs = []
    for fused in (True, False):
        heads._FUSED_OFFSETS = fused
        try:
            p = pred.detach().clone().requires_grad_()
            out = _Head().offset_to_pts(center_list, [p], y_first=y_first)[0]
            out.backward(g)
            outs.append((out.detach(), p.grad))
        finally:
            heads._FUSED_OFFSETS = True
    assert outs[0][0].shape == (B, H * W, C)
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize('channels_last', [False, True])
@pytest.mark.parametrize('shape', [(2, 256, 25, 42), (3, 256, 100, 168), (32, 256, 100, 168)])
@pytest.mark.parametrize('relu', [True, False])
def test_bf16_group_norm_relu_equals_autocast_chain(relu, channels_last, shape):
    """inference under autocast: GroupNorm (+ ReLU) reading and writing bf16 (csrc/group_norm.hip gn_act_forward<bf16>) against
    torch's chain -- bf16 -> fp32 cast, fp32 group_norm, ReLU, fp32 -> bf16 cast: the same tensor up to one bf16 ulp where the fp32
    results differ in rounding"""
    import torch.nn.functional as F
    from kgdet_amd import layers
    torch.manual_seed(3)
    gn = torch.nn.GroupNorm(32, 256).cuda()
    gn.weight.data.normal_(1.0, 0.5)
    gn.bias.data.normal_(0, 0.5)
    # (100 x 168: 8 channels x 16800 pixels per group -- the split kernels, kgdet_gn_act_forward_bf16_split; 32 images x 32 groups
    # = 1024 (image, group) pairs: the slice count must not fall back to ONE slice, which the one-workgroup kernel rejects)
    x = (torch.randn(*shape, device='cuda') * 3 + 1).bfloat16()
    if channels_last:
        x = x.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        assert layers.gn_act_bf16_applicable(x, gn)
        got = layers.gn_act_bf16(x, gn, relu)
        want = F.group_norm(x.float(), 32, gn.weight, gn.bias, gn.eps)
        if relu:
            want = want.relu()
        want_b = want.bfloat16()
    assert got.dtype == torch.bfloat16 and got.shape == x.shape and got.stride() == x.stride()    # the input's layout
    diff = (got.float() - want_b.float()).abs()
    ulp = want.abs().clamp(min=2.0 ** -10) * 2.0 ** -7     # one bf16 step at the value's magnitude
    assert (diff <= ulp).all()
    assert (diff > 0).float().mean().item() < 0.01         # and almost everywhere the very same bf16 value


def test_conv_infer_keeps_cast_copies_and_follows_weight_updates():
    """inference under autocast: conv1x1.conv_infer == the module under autocast (NCHW 3x3 / channels-last 3x3 / 1x1 as a batched GEMM),
    the reduced-precision copies are kept across calls and rebuilt when a parameter changes"""
    from kgdet_amd import conv1x1
    torch.manual_seed(0)
    for k, cl in ((3, False), (3, True), (1, False)):
        conv = torch.nn.Conv2d(64, 48, k, 1, k // 2).cuda()
        x = torch.randn(2, 64, 13, 21, device='cuda').bfloat16()
        if cl:
            x = x.contiguous(memory_format=torch.channels_last)
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
            want = conv(x)
            got = conv1x1.conv_infer(conv, x)
            cache = conv1x1._cast_cache[conv]
            assert '_kgdet_cast_cache' not in conv.__dict__             # (not deep-copied / pickled with the module)
            assert conv1x1.conv_infer(conv, x) is not None and conv1x1._cast_cache[conv] is cache      # kept
            assert got.dtype == torch.bfloat16 and got.shape == want.shape
            assert (got.float() - want.float()).abs().max().item() <= 2.0 ** -7 * want.float().abs().max().item()
            conv.weight.mul_(2.0)                                                                                # version bump
            conv.bias.add_(1.0)
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):     # (a fresh context: autocast's own cast cache is not
            got2, want2 = conv1x1.conv_infer(conv, x), conv(x)                  #  invalidated by in-place updates inside one)
            assert conv1x1._cast_cache[conv] is not cache
            assert (got2.float() - want2.float()).abs().max().item() <= 2.0 ** -7 * want2.float().abs().max().item()
        # a write THROUGH .data bumps no version counter and moves no pointer: the documented remedy is the explicit call
        # (checkpoint.load_checkpoint makes it), after which the copies follow
        with torch.no_grad():
            conv.weight.data.copy_(conv.weight.data * 0.5)
        conv1x1.invalidate_inference_caches()
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
            got3, want3 = conv1x1.conv_infer(conv, x), conv(x)
            assert (got3.float() - want3.float()).abs().max().item() <= 2.0 ** -6 * want3.float().abs().max().item()
        # with gradients enabled (training) the module itself runs
        y = conv1x1.conv_infer(conv, x.float().contiguous())
        assert y.requires_grad


def test_odd_pixel_count_maps_take_the_split_kernels_behind_a_zero_column():
    """13 x 21 / 7 x 11 levels of a five-level head: ConvModule and conv_bias_act append one zero column, run the split-operand
    kernels and cut the column off again -- same values and gradients as the fp32 convolution to split-arithmetic rounding"""
    import torch.nn.functional as F
    from kgdet_amd import conv1x1, layers
    torch.manual_seed(1)
    for H, W in ((13, 21), (7, 11)):
        x = torch.randn(2, 64, H, W, device='cuda', requires_grad=True)
        m = layers.ConvModule(64, 32, 3, padding=1, norm_cfg=dict(type='GN', num_groups=4, requires_grad=True)).cuda()
        assert conv1x1.odd_map_applicable(x, m.conv.weight, (1, 1), (1, 1), (1, 1), 1)
        conv = torch.nn.Conv2d(64, 32, 3, 1, 1).cuda()
        gy = torch.randn(2, 32, H, W, device='cuda')
        outs = []
        for flag in (True, False):
            conv1x1.PAD_ODD_MAPS = flag
            try:
                xa = x.detach().clone().requires_grad_()
                ya = m(xa) + conv1x1.conv_bias_act(conv, xa, relu=True)
                ya.backward(gy)
                outs.append((ya.detach(), xa.grad, m.conv.weight.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone()))
                m.zero_grad(); conv.zero_grad()
            finally:
                conv1x1.PAD_ODD_MAPS = True
        for a, b in zip(*outs):
            assert a.shape == b.shape and a.is_contiguous()
            assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item() + 1e-7
