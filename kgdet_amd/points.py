"""Point grid, GT assignment and training targets for the keypoint-guided heads.

Host-side mirror (small tensors, no custom kernels) of:
  PointGenerator ............ mmdet/core/anchor/point_generator.py:4-34
  PointAssigner ............. mmdet/core/bbox/assigners/point_assigner.py:7-121
  MaxIoUAssigner ............ mmdet/core/bbox/assigners/max_iou_assigner.py:7-153
  bbox_overlaps ............. mmdet/core/bbox/geometry.py:4-63
  AssignResult .............. mmdet/core/bbox/assigners/assign_result.py:4-19
  PseudoSamplerKp ........... mmdet/core/bbox/samplers/pseudo_sampler_kp.py:7-27
  SamplingResultKp .......... mmdet/core/bbox/samplers/sampling_result_kp.py:4-25
  point_target_kp (+single) . mmdet/core/anchor/point_target_kp.py:7-182
  multi_apply ............... mmdet/core/utils/misc.py
Semantics (including GT-order dependence of PointAssigner and the in-level top-k) are kept;
the only deliberate difference is that every function takes its device from its inputs instead of
defaulting to 'cuda', so the same code is testable on CPU.
"""
from functools import partial

import torch


def multi_apply(func, *args, **kwargs):
    pfunc = partial(func, **kwargs) if kwargs else func
    map_results = map(pfunc, *args)
    return tuple(map(list, zip(*map_results)))


class PointGenerator(object):

    def _meshgrid(self, x, y, row_major=True):
        xx = x.repeat(len(y))
        yy = y.view(-1, 1).repeat(1, len(x)).view(-1)
        return (xx, yy) if row_major else (yy, xx)

    def grid_points(self, featmap_size, stride=16, device='cuda'):
        """[H*W, 3] rows (x*stride, y*stride, stride), row-major, no half-stride shift."""
        feat_h, feat_w = featmap_size
        shift_x = torch.arange(0., feat_w, device=device) * stride
        shift_y = torch.arange(0., feat_h, device=device) * stride
        shift_xx, shift_yy = self._meshgrid(shift_x, shift_y)
        stride = shift_x.new_full((shift_xx.shape[0], ), stride)
        return torch.stack([shift_xx, shift_yy, stride], dim=-1)

    def valid_flags(self, featmap_size, valid_size, device='cuda'):
        feat_h, feat_w = featmap_size
        valid_h, valid_w = valid_size
        assert valid_h <= feat_h and valid_w <= feat_w
        valid_x = torch.zeros(feat_w, dtype=torch.bool, device=device)
        valid_y = torch.zeros(feat_h, dtype=torch.bool, device=device)
        valid_x[:valid_w] = 1
        valid_y[:valid_h] = 1
        valid_xx, valid_yy = self._meshgrid(valid_x, valid_y)
        return valid_xx & valid_yy


class AssignResult(object):

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts = num_gts
        self.gt_inds = gt_inds
        self.max_overlaps = max_overlaps
        self.labels = labels

    def add_gt_(self, gt_labels):
        self_inds = torch.arange(1, len(gt_labels) + 1, dtype=torch.long, device=gt_labels.device)
        self.gt_inds = torch.cat([self_inds, self.gt_inds])
        self.max_overlaps = torch.cat([self.max_overlaps.new_ones(self.num_gts), self.max_overlaps])
        if self.labels is not None:
            self.labels = torch.cat([gt_labels, self.labels])


class BaseAssigner(object):

    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None):
        raise NotImplementedError


def _labels_of(assigned_gt_inds, gt_labels, num):
    if gt_labels is None:
        return None
    assigned_labels = assigned_gt_inds.new_zeros((num, ))
    pos_inds = torch.nonzero(assigned_gt_inds > 0).squeeze()
    if pos_inds.numel() > 0:
        assigned_labels[pos_inds] = gt_labels[assigned_gt_inds[pos_inds] - 1]
    return assigned_labels


class PointAssigner(BaseAssigner):
    """0 = negative, i > 0 = positive for GT i-1.  A point is positive for a GT when it is among the
    ``pos_num`` nearest points (on the GT's pyramid level) and nearer to it than to any earlier GT."""

    def __init__(self, scale=4, pos_num=3, pos_scale_factor=None):
        self.scale = scale
        self.pos_num = pos_num
        self.pos_scale_factor = pos_scale_factor

    def assign(self, points, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None):
        if points.shape[0] == 0 or gt_bboxes.shape[0] == 0:
            raise ValueError('No gt or bboxes')
        points_xy = points[:, :2]
        points_lvl = torch.log2(points[:, 2]).int()
        lvl_min, lvl_max = points_lvl.min(), points_lvl.max()
        num_gts, num_points = gt_bboxes.shape[0], points.shape[0]

        gt_xy = (gt_bboxes[:, :2] + gt_bboxes[:, 2:]) / 2
        gt_wh = (gt_bboxes[:, 2:] - gt_bboxes[:, :2]).clamp(min=1e-6)
        gt_lvl = ((torch.log2(gt_wh[:, 0] / self.scale) + torch.log2(gt_wh[:, 1] / self.scale)) / 2).int()
        gt_lvl = torch.clamp(gt_lvl, min=lvl_min, max=lvl_max)

        assigned_gt_inds = points.new_zeros((num_points, ), dtype=torch.long)
        assigned_gt_dist = points.new_full((num_points, ), float('inf'))
        points_range = torch.arange(num_points, device=points.device)

        for idx in range(num_gts):
            on_level = gt_lvl[idx] == points_lvl
            level_index = points_range[on_level]
            dist = ((points_xy[on_level, :] - gt_xy[[idx], :]) / gt_wh[[idx], :]).norm(dim=1)
            if self.pos_scale_factor is not None:
                pos_num = (dist < self.pos_scale_factor).sum()
            else:
                pos_num = self.pos_num
            min_dist, min_dist_index = torch.topk(dist, pos_num, largest=False)
            cand = level_index[min_dist_index]
            closer = min_dist < assigned_gt_dist[cand]
            cand = cand[closer]
            assigned_gt_inds[cand] = idx + 1
            assigned_gt_dist[cand] = min_dist[closer]

        return AssignResult(num_gts, assigned_gt_inds, None,
                            labels=_labels_of(assigned_gt_inds, gt_labels, num_points))


def bbox_overlaps(bboxes1, bboxes2, mode='iou', is_aligned=False):
    """IoU / IoF with the +1 pixel convention."""
    assert mode in ['iou', 'iof']
    rows, cols = bboxes1.size(0), bboxes2.size(0)
    if is_aligned:
        assert rows == cols
    if rows * cols == 0:
        return bboxes1.new(rows, 1) if is_aligned else bboxes1.new(rows, cols)
    area1 = (bboxes1[:, 2] - bboxes1[:, 0] + 1) * (bboxes1[:, 3] - bboxes1[:, 1] + 1)
    area2 = (bboxes2[:, 2] - bboxes2[:, 0] + 1) * (bboxes2[:, 3] - bboxes2[:, 1] + 1)
    if is_aligned:
        lt = torch.max(bboxes1[:, :2], bboxes2[:, :2])
        rb = torch.min(bboxes1[:, 2:], bboxes2[:, 2:])
        wh = (rb - lt + 1).clamp(min=0)
        overlap = wh[:, 0] * wh[:, 1]
        return overlap / (area1 + area2 - overlap) if mode == 'iou' else overlap / area1
    lt = torch.max(bboxes1[:, None, :2], bboxes2[:, :2])
    rb = torch.min(bboxes1[:, None, 2:], bboxes2[:, 2:])
    wh = (rb - lt + 1).clamp(min=0)
    overlap = wh[:, :, 0] * wh[:, :, 1]
    if mode == 'iou':
        return overlap / (area1[:, None] + area2 - overlap)
    return overlap / (area1[:, None])


class MaxIoUAssigner(BaseAssigner):
    """-1 = don't care, 0 = negative, i > 0 = positive for GT i-1 (serial / parallel heads' refine stage)."""

    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=.0, gt_max_assign_all=True, ignore_iof_thr=-1,
                 ignore_wrt_candidates=True):
        self.pos_iou_thr = pos_iou_thr
        self.neg_iou_thr = neg_iou_thr
        self.min_pos_iou = min_pos_iou
        self.gt_max_assign_all = gt_max_assign_all
        self.ignore_iof_thr = ignore_iof_thr
        self.ignore_wrt_candidates = ignore_wrt_candidates

    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None):
        if bboxes.shape[0] == 0 or gt_bboxes.shape[0] == 0:
            raise ValueError('No gt or bboxes')
        bboxes = bboxes[:, :4]
        overlaps = bbox_overlaps(gt_bboxes, bboxes)
        if (self.ignore_iof_thr > 0) and (gt_bboxes_ignore is not None) and (gt_bboxes_ignore.numel() > 0):
            if self.ignore_wrt_candidates:
                ignore_max_overlaps, _ = bbox_overlaps(bboxes, gt_bboxes_ignore, mode='iof').max(dim=1)
            else:
                ignore_max_overlaps, _ = bbox_overlaps(gt_bboxes_ignore, bboxes, mode='iof').max(dim=0)
            overlaps[:, ignore_max_overlaps > self.ignore_iof_thr] = -1
        return self.assign_wrt_overlaps(overlaps, gt_labels)

    def assign_wrt_overlaps(self, overlaps, gt_labels=None):
        if overlaps.numel() == 0:
            raise ValueError('No gt or proposals')
        num_gts, num_bboxes = overlaps.size(0), overlaps.size(1)
        assigned_gt_inds = overlaps.new_full((num_bboxes, ), -1, dtype=torch.long)
        max_overlaps, argmax_overlaps = overlaps.max(dim=0)
        gt_max_overlaps, gt_argmax_overlaps = overlaps.max(dim=1)

        if isinstance(self.neg_iou_thr, float):
            assigned_gt_inds[(max_overlaps >= 0) & (max_overlaps < self.neg_iou_thr)] = 0
        elif isinstance(self.neg_iou_thr, tuple):
            assert len(self.neg_iou_thr) == 2
            assigned_gt_inds[(max_overlaps >= self.neg_iou_thr[0]) & (max_overlaps < self.neg_iou_thr[1])] = 0

        pos_inds = max_overlaps >= self.pos_iou_thr
        assigned_gt_inds[pos_inds] = argmax_overlaps[pos_inds] + 1

        for i in range(num_gts):
            if gt_max_overlaps[i] >= self.min_pos_iou:
                if self.gt_max_assign_all:
                    assigned_gt_inds[overlaps[i, :] == gt_max_overlaps[i]] = i + 1
                else:
                    assigned_gt_inds[gt_argmax_overlaps[i]] = i + 1

        return AssignResult(num_gts, assigned_gt_inds, max_overlaps,
                            labels=_labels_of(assigned_gt_inds, gt_labels, num_bboxes))


_ASSIGNERS = {'PointAssigner': PointAssigner, 'MaxIoUAssigner': MaxIoUAssigner}


def build_assigner(cfg, **kwargs):
    """``dict(type='PointAssigner', ...)`` -> assigner (mmdet/core/bbox/assign_sampling.py:6-13)."""
    if isinstance(cfg, BaseAssigner):
        return cfg
    if isinstance(cfg, dict):
        args = dict(cfg)
        name = args.pop('type')
        if name not in _ASSIGNERS:
            raise AttributeError("module 'assigners' has no attribute '{}'".format(name))
        for k, v in kwargs.items():
            args.setdefault(k, v)
        return _ASSIGNERS[name](**args)
    raise TypeError('Invalid type {} for building a sampler'.format(type(cfg)))


class SamplingResultKp(object):

    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, gt_keypoints, assign_result, gt_flags):
        self.pos_inds = pos_inds
        self.neg_inds = neg_inds
        self.pos_bboxes = bboxes[pos_inds]
        self.neg_bboxes = bboxes[neg_inds]
        self.pos_is_gt = gt_flags[pos_inds]
        self.num_gts = gt_bboxes.shape[0]
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        self.pos_gt_bboxes = gt_bboxes[self.pos_assigned_gt_inds, :]
        self.pos_gt_keypoints = gt_keypoints[self.pos_assigned_gt_inds, :]
        self.pos_gt_labels = assign_result.labels[pos_inds] if assign_result.labels is not None else None

    @property
    def bboxes(self):
        return torch.cat([self.pos_bboxes, self.neg_bboxes])


class PseudoSamplerKp(object):
    """no sampling: every positive / negative point is used"""

    def __init__(self, **kwargs):
        pass

    def sample(self, assign_result, bboxes, gt_bboxes, gt_keypoints, **kwargs):
        pos_inds = torch.nonzero(assign_result.gt_inds > 0).squeeze(-1).unique()
        neg_inds = torch.nonzero(assign_result.gt_inds == 0).squeeze(-1).unique()
        gt_flags = bboxes.new_zeros(bboxes.shape[0], dtype=torch.uint8)
        return SamplingResultKp(pos_inds, neg_inds, bboxes, gt_bboxes, gt_keypoints, assign_result, gt_flags)


def unmap(data, count, inds, fill=0):
    """scatter a subset back into a tensor of ``count`` rows"""
    if data.dim() == 1:
        ret = data.new_full((count, ), fill)
        ret[inds] = data
    else:
        ret = data.new_full((count, ) + data.size()[1:], fill)
        ret[inds, :] = data
    return ret


def images_to_levels(target, num_level_grids):
    """[target_img0, target_img1] -> [target_level0, target_level1, ...]"""
    target = torch.stack(target, 0)
    level_targets = []
    start = 0
    for n in num_level_grids:
        end = start + n
        level_targets.append(target[:, start:end].squeeze(0))
        start = end
    return level_targets


def point_target_single(flat_proposals, valid_flags, gt_bboxes, gt_keypoints, gt_bboxes_ignore, gt_labels, cfg,
                        label_channels=1, sampling=True, unmap_outputs=True):
    inside_flags = valid_flags.bool()
    if not inside_flags.any():
        return (None, ) * 9
    proposals = flat_proposals[inside_flags, :]

    if sampling:
        raise NotImplementedError('random samplers are not part of the KGDet path (focal loss => sampling=False)')
    bbox_assigner = build_assigner(cfg.assigner)
    assign_result = bbox_assigner.assign(proposals, gt_bboxes, gt_bboxes_ignore, gt_labels)
    sampling_result = PseudoSamplerKp().sample(assign_result, proposals, gt_bboxes, gt_keypoints)

    n = proposals.shape[0]
    bbox_gt = proposals.new_zeros([n, 4])
    pos_proposals = torch.zeros_like(proposals)
    proposals_weights = proposals.new_zeros([n, 4])
    labels = proposals.new_zeros(n, dtype=torch.long)
    label_weights = proposals.new_zeros(n, dtype=torch.float)
    keypoint_gt = proposals.new_zeros([n, gt_keypoints.size(1), 2])
    keypoint_weights = proposals.new_zeros([n, gt_keypoints.size(1), 2])

    pos_inds = sampling_result.pos_inds
    neg_inds = sampling_result.neg_inds
    if len(pos_inds) > 0:
        bbox_gt[pos_inds, :] = sampling_result.pos_gt_bboxes
        pos_proposals[pos_inds, :] = proposals[pos_inds, :]
        proposals_weights[pos_inds, :] = 1.0
        pos_gt_keypoints = sampling_result.pos_gt_keypoints
        keypoint_gt[pos_inds, :] = pos_gt_keypoints[:, :, :2]
        keypoint_weights[pos_inds, :] = (pos_gt_keypoints[:, :, 2:3] != 0).float()
        if gt_labels is None:
            labels[pos_inds] = 1
        else:
            labels[pos_inds] = gt_labels[sampling_result.pos_assigned_gt_inds]
        label_weights[pos_inds] = 1.0 if cfg.pos_weight <= 0 else cfg.pos_weight
    if len(neg_inds) > 0:
        label_weights[neg_inds] = 1.0

    if unmap_outputs:
        total = flat_proposals.size(0)
        labels = unmap(labels, total, inside_flags)
        label_weights = unmap(label_weights, total, inside_flags)
        bbox_gt = unmap(bbox_gt, total, inside_flags)
        pos_proposals = unmap(pos_proposals, total, inside_flags)
        proposals_weights = unmap(proposals_weights, total, inside_flags)
        keypoint_gt = unmap(keypoint_gt, total, inside_flags)
        keypoint_weights = unmap(keypoint_weights, total, inside_flags)
    return (labels, label_weights, bbox_gt, pos_proposals, proposals_weights, keypoint_gt, keypoint_weights,
            pos_inds, neg_inds)


# ------------------------------------------------------------------------------------------------
# Sync-free targets for the KGDet configuration.
# The mirrored path above selects positives with nonzero() / boolean-mask indexing: ~40 device->host
# round trips per step, each of which stalls the launch queue (measured: 4.8 ms of an idle GPU per 33 ms
# training step).  When every point lies on ONE pyramid level (KGDet: stride 32 only), is valid, and the
# assigner is a PointAssigner with a fixed pos_num, the same targets are computed with fixed-shape masked
# ops: identical values (tests/test_host_logic.py::test_dense_targets_equal_mirrored_path), no host syncs,
# and the positive count stays a device tensor.
# ------------------------------------------------------------------------------------------------
def dense_targets_applicable(cfg, num_levels, all_valid, gt_bboxes_ignore_list=None):
    """any number of pyramid levels; PointAssigner with a fixed pos_num, or MaxIoUAssigner without ignore regions"""
    a = cfg.assigner
    if not all_valid or not (gt_bboxes_ignore_list is None or all(g is None for g in gt_bboxes_ignore_list)):
        return False
    if a['type'] == 'PointAssigner':
        return a.get('pos_scale_factor') is None
    return a['type'] == 'MaxIoUAssigner'


def _point_assign_dense(points, gt_bboxes, scale, pos_num):
    """PointAssigner.assign (point_assigner.py:23-121) without masked subsets: the distance of every point to every gt
    (inf off the gt's pyramid level), the pos_num nearest per gt by ONE batched topk, and the sequential
    ``min_dist < assigned_dist`` rule (an earlier gt keeps a tie) as ONE first-minimum over the gts."""
    points_xy = points[:, :2]
    points_lvl = torch.log2(points[:, 2]).int()
    lvl_min, lvl_max = points_lvl.min(), points_lvl.max()
    gt_xy = (gt_bboxes[:, :2] + gt_bboxes[:, 2:]) / 2
    gt_wh = (gt_bboxes[:, 2:] - gt_bboxes[:, :2]).clamp(min=1e-6)
    gt_lvl = ((torch.log2(gt_wh[:, 0] / scale) + torch.log2(gt_wh[:, 1] / scale)) / 2).int()
    gt_lvl = torch.min(torch.max(gt_lvl, lvl_min), lvl_max)
    dist = ((points_xy[None, :, :] - gt_xy[:, None, :]) / gt_wh[:, None, :]).norm(dim=2)            # [G, N]
    dist = torch.where(gt_lvl[:, None] == points_lvl[None, :], dist, dist.new_full((), float('inf')))
    min_dist, cand = torch.topk(dist, pos_num, dim=1, largest=False)
    picked = torch.full_like(dist, float('inf')).scatter_(1, cand, min_dist)
    best, gt_of = picked.min(dim=0)                      # ties: the first (earliest) gt
    return torch.where(best < float('inf'), gt_of + 1, torch.zeros_like(gt_of))


def _max_iou_assign_dense(bboxes, gt_bboxes, a):
    """MaxIoUAssigner.assign_wrt_overlaps (max_iou_assigner.py:93-153) with masks instead of index writes and without the
    per-gt ``if gt_max_overlaps[i] >= min_pos_iou`` host read: -1 don't care, 0 negative, i > 0 positive for gt i - 1."""
    overlaps = bbox_overlaps(gt_bboxes, bboxes[:, :4])                                                # [G, N]
    num_gts = overlaps.shape[0]
    max_overlaps, argmax_overlaps = overlaps.max(dim=0)
    gt_max_overlaps, gt_argmax_overlaps = overlaps.max(dim=1)
    assigned = torch.full_like(argmax_overlaps, -1)
    neg = a['neg_iou_thr']
    lo, hi = (0, neg) if isinstance(neg, float) else (neg[0], neg[1])
    assigned = torch.where((max_overlaps >= lo) & (max_overlaps < hi), torch.zeros_like(assigned), assigned)
    assigned = torch.where(max_overlaps >= a['pos_iou_thr'], argmax_overlaps + 1, assigned)
    ok = (gt_max_overlaps >= a.get('min_pos_iou', .0))[:, None]
    if a.get('gt_max_assign_all', True):
        hit = (overlaps == gt_max_overlaps[:, None]) & ok
    else:
        hit = (torch.arange(overlaps.shape[1], device=overlaps.device)[None, :] == gt_argmax_overlaps[:, None]) & ok
    # the reference loops over the gts in order: the LAST gt that claims a box wins
    last = (hit.long() * torch.arange(1, num_gts + 1, device=overlaps.device)[:, None]).max(dim=0)[0]
    return torch.where(last > 0, last, assigned)


def point_target_kp_dense(proposals_list, gt_bboxes_list, gt_kps_list, cfg, gt_labels_list=None):
    """Same return value as point_target_kp(..., sampling=False) for all-valid point sets (any number of levels);
    num_total_pos / num_total_neg are 0-dim device tensors."""
    a = cfg.assigner
    pos_weight = 1.0 if cfg.pos_weight <= 0 else cfg.pos_weight
    outs = [[] for _ in range(7)]
    num_total_pos = num_total_neg = None
    first = proposals_list[0]
    num_level = [p.shape[0] for p in first] if isinstance(first, (list, tuple)) else [first.shape[0]]
    for i, proposals in enumerate(proposals_list):
        proposals = torch.cat(proposals) if isinstance(proposals, (list, tuple)) else proposals
        gt_bboxes, gt_kps = gt_bboxes_list[i], gt_kps_list[i]
        if proposals.shape[0] == 0 or gt_bboxes.shape[0] == 0:
            raise ValueError('No gt or bboxes')
        if a['type'] == 'PointAssigner':
            inds = _point_assign_dense(proposals, gt_bboxes, a.get('scale', 4), a.get('pos_num', 3))
        else:
            inds = _max_iou_assign_dense(proposals, gt_bboxes, a)
        pos = inds > 0
        gidx = (inds - 1).clamp(min=0)
        pos1, pos2 = pos[:, None], pos[:, None, None]
        kp = gt_kps[gidx]                                                   # [P, n_kp, 3]
        labels = (torch.ones_like(inds) if gt_labels_list is None or gt_labels_list[i] is None
                  else gt_labels_list[i][gidx])
        outs[0].append(torch.where(pos, labels, torch.zeros_like(labels)))
        # label weights: positives pos_weight, negatives (assigned == 0) 1, don't-care (-1, MaxIoUAssigner) 0
        lw = torch.where(pos, proposals.new_full((), pos_weight), (inds == 0).to(proposals.dtype))
        outs[1].append(lw.contiguous())
        outs[2].append(torch.where(pos1, gt_bboxes[gidx], gt_bboxes.new_zeros(())))
        outs[3].append(torch.where(pos1, proposals, proposals.new_zeros(())))
        outs[4].append(pos1.to(proposals.dtype).expand(-1, 4).contiguous())
        outs[5].append(torch.where(pos2, kp[:, :, :2], kp.new_zeros(())))
        outs[6].append(torch.where(pos2, (kp[:, :, 2:3] != 0).to(proposals.dtype), kp.new_zeros(()))
                       .expand(-1, -1, 2).contiguous())
        n_pos = pos.sum()
        n_neg = (inds == 0).sum()
        num_total_pos = n_pos.clamp(min=1) if num_total_pos is None else num_total_pos + n_pos.clamp(min=1)
        num_total_neg = n_neg.clamp(min=1) if num_total_neg is None else num_total_neg + n_neg.clamp(min=1)
    return tuple(images_to_levels(o, num_level) for o in outs) + (num_total_pos, num_total_neg)


def point_target_kp(proposals_list, valid_flag_list, gt_bboxes_list, gt_kps_list, img_metas, cfg,
                    gt_bboxes_ignore_list=None, gt_labels_list=None, label_channels=1, sampling=True,
                    unmap_outputs=True):
    """Targets of all images, regrouped per pyramid level.  Returns ``None`` if an image has no valid point."""
    num_imgs = len(img_metas)
    assert len(proposals_list) == len(valid_flag_list) == num_imgs
    num_level_proposals = [points.size(0) for points in proposals_list[0]]
    for i in range(num_imgs):
        assert len(proposals_list[i]) == len(valid_flag_list[i])
        proposals_list[i] = torch.cat(proposals_list[i])
        valid_flag_list[i] = torch.cat(valid_flag_list[i])

    if gt_bboxes_ignore_list is None:
        gt_bboxes_ignore_list = [None for _ in range(num_imgs)]
    if gt_labels_list is None:
        gt_labels_list = [None for _ in range(num_imgs)]
    (all_labels, all_label_weights, all_bbox_gt, all_proposals, all_proposal_weights, all_keypoint_gt,
     all_keypoint_weights, pos_inds_list, neg_inds_list) = multi_apply(
         point_target_single, proposals_list, valid_flag_list, gt_bboxes_list, gt_kps_list, gt_bboxes_ignore_list,
         gt_labels_list, cfg=cfg, label_channels=label_channels, sampling=sampling, unmap_outputs=unmap_outputs)
    if any([labels is None for labels in all_labels]):
        return None
    num_total_pos = sum([max(inds.numel(), 1) for inds in pos_inds_list])
    num_total_neg = sum([max(inds.numel(), 1) for inds in neg_inds_list])
    return (images_to_levels(all_labels, num_level_proposals),
            images_to_levels(all_label_weights, num_level_proposals),
            images_to_levels(all_bbox_gt, num_level_proposals),
            images_to_levels(all_proposals, num_level_proposals),
            images_to_levels(all_proposal_weights, num_level_proposals),
            images_to_levels(all_keypoint_gt, num_level_proposals),
            images_to_levels(all_keypoint_weights, num_level_proposals), num_total_pos, num_total_neg)
