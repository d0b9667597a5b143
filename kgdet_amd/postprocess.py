"""Per-class NMS of decoded detections, keypoints carried along.

``multiclass_nms_kp`` mirrors mmdet/core/post_processing/bbox_nms_kp.py:6-75 call for call (the NMS
type is looked up by name exactly as ``getattr(nms_wrapper, nms_type)`` does there).
``multiclass_nms_kp_batched`` is the MI355X path for hard NMS: every (image, class) group of the
whole batch goes through ONE launch of the batched HIP NMS, with two small host reads per batch
instead of one synchronising NMS call per class per image.  Both return, per image,
(det_bboxes [k,5], det_labels [k], det_kpts [k, ...]) in the reference's order: class-major,
ascending candidate index inside a class, re-sorted by score only when more than ``max_num`` survive.
"""
import torch

from . import nms as nms_wrapper


def multiclass_nms_kp(multi_bboxes, multi_scores, multi_kpts, score_thr, nms_cfg, max_num=-1, score_factors=None):
    """One image: per foreground class, candidates with ``score > score_thr`` go through the NMS op named by
    ``nms_cfg['type']`` (boxes [N, 4] shared by the classes or [N, 4 * classes]); survivors are concatenated class by
    class, and when more than ``max_num`` remain the top ``max_num`` by score are kept (bbox_nms_kp.py:6-75, including its
    ``inds[:max_num]`` slice for ``max_num = -1``).  The keypoints of a survivor follow it."""
    assert multi_kpts.shape[1] % 3 == 0
    op_args = dict(nms_cfg)
    op = getattr(nms_wrapper, op_args.pop('type', 'nms'))
    shared_boxes = multi_bboxes.shape[1] == 4
    candidate = multi_scores > score_thr                        # column 0 (background) is never looked at
    dets, labels, rows = [], [], []
    for cls in range(1, multi_scores.shape[1]):
        picked = candidate[:, cls]
        if not bool(picked.any()):
            continue
        boxes = multi_bboxes[picked] if shared_boxes else multi_bboxes[picked, 4 * cls:4 * cls + 4]
        score = multi_scores[picked, cls]
        if score_factors is not None:
            score = score * score_factors[picked]
        kept_dets, kept = op(torch.cat([boxes, score.unsqueeze(1)], dim=1), **op_args)
        dets.append(kept_dets)
        labels.append(torch.full((kept_dets.shape[0], ), cls - 1, dtype=torch.long, device=multi_bboxes.device))
        rows.append(picked.nonzero().flatten()[kept])
    if not dets:
        return (multi_bboxes.new_zeros((0, 5)), multi_bboxes.new_zeros((0, ), dtype=torch.long),
                multi_bboxes.new_zeros((0, multi_kpts.shape[1])))
    dets, labels, kpts = torch.cat(dets), torch.cat(labels), multi_kpts[torch.cat(rows)]
    if dets.shape[0] > max_num:
        top = dets[:, 4].sort(descending=True)[1][:max_num]
        dets, labels, kpts = dets[top], labels[top], kpts[top]
    return dets, labels, kpts


def multiclass_nms_kp_batched(bboxes, scores, kpts, score_thr, nms_cfg, max_num=-1):
    """bboxes [B,N,4], scores [B,N,1+C] (column 0 = background), kpts [B,N,...] on the GPU.

    Returns a list of B (det_bboxes, det_labels, det_kpts) tuples identical to calling
    ``multiclass_nms_kp`` per image with ``type='nms'``.
    """
    nms_cfg_ = dict(nms_cfg)
    nms_type = nms_cfg_.pop('type', 'nms')
    if nms_type != 'nms':
        return [multiclass_nms_kp(bboxes[b], scores[b], kpts[b].reshape(kpts.shape[1], -1), score_thr, nms_cfg,
                                  max_num) for b in range(bboxes.shape[0])]
    iou_thr = float(nms_cfg_['iou_thr'])
    B, N = scores.shape[0], scores.shape[1]
    C = scores.shape[2] - 1
    cand = (scores[:, :, 1:] > score_thr).permute(0, 2, 1).contiguous()   # [B, C, N] -> row-major = (b, c, n)
    counts = cand.sum(-1).flatten()                                        # [B*C]
    offsets = torch.zeros(B * C + 1, dtype=torch.int64, device=scores.device)
    offsets[1:] = counts.cumsum(0)
    idx = cand.nonzero()                                                   # [T, 3] sorted by (b, c, n)  (host read #1)
    T = idx.shape[0]
    empty = (bboxes.new_zeros((0, 5)), bboxes.new_zeros((0, ), dtype=torch.long),
             bboxes.new_zeros((0, ) + tuple(kpts.shape[2:])))
    if T == 0:
        return [empty for _ in range(B)]
    b_i, c_i, n_i = idx[:, 0], idx[:, 1], idx[:, 2]
    dets = torch.cat([bboxes[b_i, n_i], scores[b_i, n_i, c_i + 1].unsqueeze(1)], 1).contiguous()
    keep, num_keep = nms_wrapper.nms_batched(dets, offsets, iou_thr, max_seg_len=N)
    # kept entries -> flat positions, still ordered by (b, c, ascending candidate index)
    seg_of = torch.repeat_interleave(torch.arange(B * C, device=dets.device), counts, output_size=T)
    pos_in_seg = torch.arange(T, device=dets.device) - offsets[seg_of]
    is_kept_slot = pos_in_seg < num_keep[seg_of]
    kept_flat = (offsets[seg_of] + keep)[is_kept_slot]                     # (host read #2: size)
    kept_img = seg_of[is_kept_slot] // C
    per_img = torch.bincount(kept_img, minlength=B).tolist()
    out, start = [], 0
    for b in range(B):
        sel = kept_flat[start:start + per_img[b]]
        start += per_img[b]
        if sel.numel() == 0:
            out.append(empty)
            continue
        d, lab, kp = dets[sel], c_i[sel], kpts[b_i[sel], n_i[sel]]
        if d.shape[0] > max_num:
            _, order = d[:, -1].sort(descending=True, stable=True)
            order = order[:max_num]
            d, lab, kp = d[order], lab[order], kp[order]
        out.append((d, lab, kp))
    return out


def multiclass_nms_kp_fused(bboxes, scores, kpts, score_thr, iou_thr, max_num):
    """``multiclass_nms_kp`` (type='nms') for a batch with nothing read by the host: two HIP launches
    (csrc/nms.hip ``multiclass_nms_segments`` + ``multiclass_select``) and one landmark gather.

    bboxes [B,N,4], scores [B,N,C] (foreground classes only), kpts [B,N,K] float32 on the GPU.
    Returns fixed-size device tensors (det [B,M,5], labels [B,M] int64 0-based, kpts [B,M,K], count [B] int64),
    rows past ``count[b]`` zero; ``M = max_num``.  Raises NotImplementedError beyond the on-chip limits
    (N <= 4096, N*C <= 16384) -- callers fall back to ``multiclass_nms_kp_batched``."""
    import ctypes
    from . import _lib
    B, N, C = scores.shape
    bboxes, scores = bboxes.contiguous().float(), scores.contiguous().float()
    L = _lib.lib()
    L.kgdet_multiclass_nms_workspace_bytes.restype = ctypes.c_size_t
    ws_bytes = L.kgdet_multiclass_nms_workspace_bytes(ctypes.c_int32(B), ctypes.c_int32(N), ctypes.c_int32(C))
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=bboxes.device)
    det = torch.empty((B, max_num, 5), dtype=torch.float32, device=bboxes.device)
    label = torch.empty((B, max_num), dtype=torch.int64, device=bboxes.device)
    src = torch.empty((B, max_num), dtype=torch.int64, device=bboxes.device)
    count = torch.empty((B, ), dtype=torch.int64, device=bboxes.device)
    _lib.check(L.kgdet_multiclass_nms(
        _lib.ptr(bboxes), _lib.ptr(scores), ctypes.c_int32(B), ctypes.c_int32(N), ctypes.c_int32(C), ctypes.c_int32(C),
        ctypes.c_int32(0), ctypes.c_float(score_thr), ctypes.c_float(iou_thr), ctypes.c_int32(max_num), _lib.ptr(det),
        _lib.ptr(label), _lib.ptr(src), _lib.ptr(count), _lib.ptr(ws), ctypes.c_size_t(ws_bytes),
        _lib.current_stream()), 'kgdet_multiclass_nms')
    k = kpts.reshape(B, N, -1)
    out_k = torch.gather(k, 1, src.unsqueeze(-1).expand(B, max_num, k.shape[-1]))
    out_k = out_k * (torch.arange(max_num, device=count.device).unsqueeze(0) < count.unsqueeze(1)).unsqueeze(-1)
    return det, label, out_k, count


_SOFT_METHODS = {'linear': 1, 'gaussian': 2}      # (nms_wrapper.py:66-68; anything else is the reference's ValueError)


def soft_nms_fused_supported(B, N, C, max_num):
    """the library's own statement of the fused soft-NMS limits (csrc/nms.hip kgdet_multiclass_soft_nms_supported)"""
    from . import _lib
    return bool(_lib.lib().kgdet_multiclass_soft_nms_supported(int(B), int(N), int(C), int(max_num)))


def multiclass_soft_nms_kp_fused(bboxes, scores, kpts, score_thr, nms_cfg, max_num):
    """``multiclass_nms_kp`` with ``nms_cfg['type'] == 'soft_nms'`` for a batch with nothing read by the host: two HIP launches
    (csrc/nms.hip ``multiclass_soft_nms_segments`` + ``multiclass_soft_select``) and one landmark gather -- what lets the
    config-5 (serial head, soft-NMS) inference batch be captured as one HIP graph.  Same arguments and returns as
    ``multiclass_nms_kp_fused``; the fifth detection column is the DECAYED score, as ``soft_nms`` returns it.  Raises
    NotImplementedError beyond the on-chip limits (N * 36 bytes of LDS, classes * max_num <= 16384)."""
    import ctypes
    from . import _lib
    cfg = dict(nms_cfg)
    assert cfg.pop('type') == 'soft_nms'
    method = cfg.get('method', 'linear')
    if method not in _SOFT_METHODS:
        raise ValueError('Invalid method for SoftNMS: {}'.format(method))
    iou_thr, sigma, min_score = float(cfg['iou_thr']), float(cfg.get('sigma', 0.5)), float(cfg.get('min_score', 1e-3))
    B, N, C = scores.shape
    if not soft_nms_fused_supported(B, N, C, max_num):
        raise NotImplementedError('beyond the fused soft-NMS limits')
    bboxes, scores = bboxes.contiguous().float(), scores.contiguous().float()
    L = _lib.lib()           # (restype of the size query: set once at library load, kgdet_amd/_lib.py)
    ws_bytes = L.kgdet_multiclass_soft_nms_workspace_bytes(ctypes.c_int32(B), ctypes.c_int32(N), ctypes.c_int32(C))
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=bboxes.device)
    det = torch.empty((B, max_num, 5), dtype=torch.float32, device=bboxes.device)
    label = torch.empty((B, max_num), dtype=torch.int64, device=bboxes.device)
    src = torch.empty((B, max_num), dtype=torch.int64, device=bboxes.device)
    count = torch.empty((B, ), dtype=torch.int64, device=bboxes.device)
    _lib.check(L.kgdet_multiclass_soft_nms(
        _lib.ptr(bboxes), _lib.ptr(scores), ctypes.c_int32(B), ctypes.c_int32(N), ctypes.c_int32(C), ctypes.c_int32(C),
        ctypes.c_int32(0), ctypes.c_float(score_thr), ctypes.c_float(iou_thr), ctypes.c_int32(_SOFT_METHODS[method]),
        ctypes.c_float(sigma), ctypes.c_float(min_score), ctypes.c_int32(max_num), _lib.ptr(det), _lib.ptr(label),
        _lib.ptr(src), _lib.ptr(count), _lib.ptr(ws), ctypes.c_size_t(ws_bytes), _lib.current_stream()),
        'kgdet_multiclass_soft_nms')
    k = kpts.reshape(B, N, -1)
    out_k = torch.gather(k, 1, src.unsqueeze(-1).expand(B, max_num, k.shape[-1]))
    out_k = out_k * (torch.arange(max_num, device=count.device).unsqueeze(0) < count.unsqueeze(1)).unsqueeze(-1)
    return det, label, out_k, count
