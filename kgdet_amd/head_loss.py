"""Target assignment + the nine losses of the KGDet head as four HIP launches (csrc/head_loss.hip).

``RepPointsHeadKp3RepCas1AssignOnce.loss`` (reppoints_head_kp3rep_cas_1_assign_once.py:581-665) builds its targets with
``PointAssigner`` / ``point_target_kp`` and evaluates 3 x (FocalLoss, SmoothL1Loss boxes, SmoothL1Loss keypoints) on
decoded coordinates.  ``kgdet_amd.heads`` keeps that path, expression by expression (``points.point_target_kp_dense``,
``losses``): ~650 small torch launches per step.  For the configuration the KGDet configs train -- one pyramid level, all
points valid, PointAssigner with a fixed ``pos_num``, sigmoid focal + smooth-L1 losses with ``reduction='mean'`` -- this
module computes the same nine numbers and the same nine gradients from the raw prediction maps and the ground-truth
tables, without materialising a target or weight tensor and without a host read.

``KGDET_FUSED_HEAD_LOSS=0`` selects the torch chain (A/B, and what the parity tests compare against).
"""
import ctypes
import os

import torch

from . import _lib

ENABLED = os.environ.get('KGDET_FUSED_HEAD_LOSS', '1') == '1'
MAX_IMAGES, MAX_GT, MAX_POINTS = 16, 64, 4096


class HeadTargets(ctypes.Structure):
    _fields_ = [('B', ctypes.c_int32), ('H', ctypes.c_int32), ('W', ctypes.c_int32), ('num_classes', ctypes.c_int32),
                ('num_keypoints', ctypes.c_int32), ('stride', ctypes.c_float),
                ('num_gt', ctypes.c_int32 * MAX_IMAGES), ('gt_bboxes', ctypes.c_void_p * MAX_IMAGES),
                ('gt_labels', ctypes.c_void_p * MAX_IMAGES), ('gt_keypoints', ctypes.c_void_p * MAX_IMAGES),
                ('valid_h', ctypes.c_int32 * MAX_IMAGES), ('valid_w', ctypes.c_int32 * MAX_IMAGES)]


class HeadLossCfg(ctypes.Structure):
    _fields_ = [('pos_num', ctypes.c_int32), ('pos_weight', ctypes.c_float), ('normalize_term', ctypes.c_float),
                ('gamma', ctypes.c_float * 3), ('alpha', ctypes.c_float * 3), ('beta', ctypes.c_float * 6),
                ('loss_weight', ctypes.c_float * 9)]


class HeadMaps(ctypes.Structure):
    _fields_ = [('cls', ctypes.c_void_p * 3), ('bbox', ctypes.c_void_p * 3), ('kpt', ctypes.c_void_p * 3)]


def _maps(tensors):
    m = HeadMaps()
    for s in range(3):
        m.cls[s], m.bbox[s], m.kpt[s] = tensors[s].data_ptr(), tensors[3 + s].data_ptr(), tensors[6 + s].data_ptr()
    return m


def applicable(head, cfg, cls_scores, kpt_preds, bbox_preds, gt_bboxes, gt_labels, gt_keypoints, gt_bboxes_ignore,
               all_valid=True, valid_sizes=None):
    """the fused kernels cover exactly the case ``points.dense_targets_applicable`` covers for one level, on the GPU, in
    float32; ``valid_sizes``: per image the (rows, columns) of the grid inside its pad_shape (None: all of it)"""
    from .losses import FocalLoss, SmoothL1Loss
    from .points import dense_targets_applicable
    if not ENABLED or head.sampling or not head.use_sigmoid_cls or len(head.point_strides) != 1:
        return False
    if not dense_targets_applicable(cfg, 1, all_valid, gt_bboxes_ignore):
        return False
    for stage in (1, 2, 3):
        lc, lb, lk = (getattr(head, 'loss_%s_%d' % (n, stage)) for n in ('cls', 'bbox', 'kpt'))
        if type(lc) is not FocalLoss or lc.reduction != 'mean' or type(lb) is not SmoothL1Loss or \
                type(lk) is not SmoothL1Loss or lb.reduction != 'mean' or lk.reduction != 'mean':
            return False
    t0 = cls_scores[0][0]
    B, _, H, W = t0.shape
    if not t0.is_cuda or B > MAX_IMAGES or H * W > MAX_POINTS or torch.is_autocast_enabled():
        return False
    for group in (cls_scores, kpt_preds, bbox_preds):
        for per_level in group:
            if len(per_level) != 1 or per_level[0].dtype != torch.float32:
                return False
    if cfg.assigner.get('pos_num', 3) > H * W:
        return False
    if valid_sizes is not None and any(vh * vw < cfg.assigner.get('pos_num', 3) for vh, vw in valid_sizes):
        return False
    for b in range(B):
        g = gt_bboxes[b].shape[0]
        if g < 1 or g > MAX_GT or gt_bboxes[b].dtype != torch.float32 or gt_keypoints[b].dtype != torch.float32 or \
                gt_keypoints[b].shape[1:] != (head.num_keypts, 3):
            return False
        # the kernels dereference the raw ground-truth pointers: they must live on the maps' device (CPU-resident or
        # other-GPU ground truth takes the torch chain, which raises torch's own device-mismatch error or works)
        if gt_bboxes[b].device != t0.device or gt_keypoints[b].device != t0.device:
            return False
        if gt_labels is not None and gt_labels[b] is not None and (gt_labels[b].dtype != torch.int64 or
                                                                   gt_labels[b].device != t0.device):
            return False
    return True


class _HeadLoss(torch.autograd.Function):
    """(cls_1..3, bbox_1..3, kpt_1..3 prediction maps) -> nine 0-dim losses"""

    @staticmethod
    def forward(ctx, targets, cfg, keep, *maps):
        L = _lib.lib()
        maps = tuple(m.contiguous() for m in maps)
        dev = maps[0].device
        ws_bytes = L.kgdet_head_loss_workspace_bytes(ctypes.byref(targets))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        out = torch.empty(10, dtype=torch.float32, device=dev)          # nine losses + num_total
        hm = _maps(maps)
        _lib.check(L.kgdet_head_loss_forward(ctypes.byref(targets), ctypes.byref(cfg), ctypes.byref(hm), _lib.ptr(out),
                                             ctypes.c_void_p(out.data_ptr() + 36), _lib.ptr(ws), ctypes.c_size_t(ws_bytes),
                                             _lib.current_stream()), 'kgdet_head_loss_forward')
        ctx.targets, ctx.cfg, ctx.keep, ctx.ws, ctx.ws_bytes, ctx.out = targets, cfg, keep, ws, ws_bytes, out
        ctx.save_for_backward(*maps)
        return tuple(out[k] for k in range(9))

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grad_losses):
        maps = ctx.saved_tensors
        dev = maps[0].device
        zero = None
        gl = []
        for g in grad_losses:
            if g is None:
                if zero is None:
                    zero = torch.zeros((), dtype=torch.float32, device=dev)
                g = zero
            gl.append(g.reshape(()).float())
        up = torch.stack(gl)
        grads = tuple(torch.empty_like(m) for m in maps)
        hm, hg = _maps(maps), _maps(grads)
        _lib.check(_lib.lib().kgdet_head_loss_backward(
            ctypes.byref(ctx.targets), ctypes.byref(ctx.cfg), ctypes.byref(hm), _lib.ptr(up),
            ctypes.c_void_p(ctx.out.data_ptr() + 36), ctypes.byref(hg), _lib.ptr(ctx.ws), ctypes.c_size_t(ctx.ws_bytes),
            _lib.current_stream()), 'kgdet_head_loss_backward')
        return (None, None, None) + grads


def head_loss(head, cfg, cls_scores, kpt_preds, bbox_preds, gt_bboxes, gt_labels, gt_keypoints, valid_sizes=None):
    """The nine losses of ``head.loss`` as a dict of 0-dim tensors (one list entry per level, as the reference returns).
    ``cls_scores`` / ``kpt_preds`` / ``bbox_preds``: three stages x [one level] of NCHW maps."""
    t0 = cls_scores[0][0]
    B, C, H, W = t0.shape
    t = HeadTargets()
    t.B, t.H, t.W, t.num_classes, t.num_keypoints = B, H, W, C, head.num_keypts
    t.stride = float(head.point_strides[0])
    keep = []                                    # the contiguous ground-truth tensors the pointers refer to
    for b in range(B):
        bb, kp = gt_bboxes[b].contiguous(), gt_keypoints[b].contiguous()
        if bb.shape[0] == 0:
            raise ValueError('No gt or bboxes')
        lab = None if gt_labels is None or gt_labels[b] is None else gt_labels[b].contiguous()
        keep += [bb, kp, lab]
        t.num_gt[b] = bb.shape[0]
        if valid_sizes is not None:
            t.valid_h[b], t.valid_w[b] = int(valid_sizes[b][0]), int(valid_sizes[b][1])
        t.gt_bboxes[b], t.gt_keypoints[b] = bb.data_ptr(), kp.data_ptr()
        t.gt_labels[b] = lab.data_ptr() if lab is not None else None
    c = HeadLossCfg()
    a = cfg.assigner
    c.pos_num = int(a.get('pos_num', 3))
    c.pos_weight = 1.0 if cfg.pos_weight <= 0 else float(cfg.pos_weight)
    c.normalize_term = float(head.point_base_scale * head.point_strides[0])
    for s in range(3):
        lc, lb, lk = (getattr(head, 'loss_%s_%d' % (n, s + 1)) for n in ('cls', 'bbox', 'kpt'))
        c.gamma[s], c.alpha[s] = float(lc.gamma), float(lc.alpha)
        c.beta[s], c.beta[3 + s] = float(lb.beta), float(lk.beta)
        c.loss_weight[s], c.loss_weight[3 + s], c.loss_weight[6 + s] = float(lc.loss_weight), float(lb.loss_weight), \
            float(lk.loss_weight)
    maps = [cls_scores[s][0] for s in range(3)] + [bbox_preds[s][0] for s in range(3)] + [kpt_preds[s][0] for s in range(3)]
    out = _HeadLoss.apply(t, c, keep, *maps)
    names = ['loss_cls_1', 'loss_cls_2', 'loss_cls_3', 'loss_bbox_1', 'loss_bbox_2', 'loss_bbox_3',
             'loss_kpt_1', 'loss_kpt_2', 'loss_kpt_3']
    return {n: [v] for n, v in zip(names, out)}
