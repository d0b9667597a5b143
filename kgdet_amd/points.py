"""Point grid, ground-truth assignment and training targets of the keypoint-guided heads.

The contract (class names, constructor keywords, the -1 / 0 / +i encoding of an assignment, the nine-tuple of
``point_target_kp``) is the reference's, because its configs and heads name these pieces:
  PointGenerator ............ mmdet/core/anchor/point_generator.py:4-34
  PointAssigner ............. mmdet/core/bbox/assigners/point_assigner.py:7-121
  MaxIoUAssigner ............ mmdet/core/bbox/assigners/max_iou_assigner.py:7-153
  bbox_overlaps ............. mmdet/core/bbox/geometry.py:4-63
  AssignResult .............. mmdet/core/bbox/assigners/assign_result.py:4-19
  PseudoSamplerKp ........... mmdet/core/bbox/samplers/pseudo_sampler_kp.py:7-27
  SamplingResultKp .......... mmdet/core/bbox/samplers/sampling_result_kp.py:4-25
  point_target_kp (+single) . mmdet/core/anchor/point_target_kp.py:7-182
The bodies are this repository's: ONE dense, fixed-shape formulation -- every point against every ground truth, masks
instead of index subsets, no nonzero() / boolean indexing, hence no device -> host read -- that also honours per-image
VALID flags (the grid points beyond an image's own ``pad_shape`` in a batch of mixed shapes: excluded from the top-k and
from the level range, label weight 0, no loss; point_target_kp.py:107-161 with ``unmap``).  The assigner classes and the
list-returning ``point_target_kp`` are front-ends of those functions; only the front-ends that must return index lists
(``pos_inds``) read the device.  Bit-exact against fixtures made by the reference's own Python, on the CPU and on the
GPU, with and without invalid points (tests/test_ref_golden.py, tests/test_gpu_ref_golden.py).
Every function takes its device from its inputs (the reference defaults to 'cuda').
"""
from functools import partial

import torch


def multi_apply(func, *args, **kwargs):
    pfunc = partial(func, **kwargs) if kwargs else func
    return tuple(map(list, zip(*map(pfunc, *args))))


# ------------------------------------------------------------------------------------------------
# grid
# ------------------------------------------------------------------------------------------------
class PointGenerator(object):

    def grid_points(self, featmap_size, stride=16, device='cuda'):
        """[H*W, 3] rows (x * stride, y * stride, stride), row-major, no half-stride shift"""
        rows, cols = featmap_size
        grid = torch.empty(rows, cols, 3, device=device)
        grid[:, :, 0] = (torch.arange(0., cols, device=device) * stride)[None, :]
        grid[:, :, 1] = (torch.arange(0., rows, device=device) * stride)[:, None]
        grid[:, :, 2] = stride
        return grid.reshape(-1, 3)

    def valid_flags(self, featmap_size, valid_size, device='cuda'):
        """[H*W] bool: the top-left ``valid_size`` block of the grid"""
        rows, cols = featmap_size
        ok_rows, ok_cols = valid_size
        assert ok_rows <= rows and ok_cols <= cols
        inside = (torch.arange(rows, device=device) < ok_rows)[:, None] & (torch.arange(cols, device=device) < ok_cols)[None, :]
        return inside.reshape(-1)


# ------------------------------------------------------------------------------------------------
# dense assignment rules
# ------------------------------------------------------------------------------------------------
def _gather_labels(inds, gt_labels):
    """label of the assigned ground truth, 0 where nothing is assigned (or None without labels)"""
    if gt_labels is None:
        return None
    picked = gt_labels[(inds - 1).clamp(min=0)]
    return torch.where(inds > 0, picked, torch.zeros_like(picked)).to(inds.dtype)


def assign_points(points, gt_bboxes, scale=4, pos_num=3, pos_scale_factor=None, valid=None):
    """PointAssigner's rule for all points at once -> [N] long, 0 = negative, i > 0 = ground truth i - 1.

    A ground truth lives on the pyramid level ``(log2(w / scale) + log2(h / scale)) / 2`` (clamped to the levels that
    have [valid] points); its candidates are the ``pos_num`` nearest [valid] points of that level in the metric
    ``|(p - centre) / size|`` (or all of them within ``pos_scale_factor``); a point claimed by several ground truths
    goes to the nearest, the earliest on a tie.  Points with ``valid == False`` are never candidates."""
    level = torch.log2(points[:, 2]).int()
    if valid is None:
        lowest, highest = level.min(), level.max()
    else:
        far = torch.iinfo(torch.int32).max
        lowest = torch.where(valid, level, torch.full_like(level, far)).min()
        highest = torch.where(valid, level, torch.full_like(level, -far)).max()
    centre = (gt_bboxes[:, :2] + gt_bboxes[:, 2:]) / 2
    size = (gt_bboxes[:, 2:] - gt_bboxes[:, :2]).clamp(min=1e-6)
    gt_level = ((torch.log2(size[:, 0] / scale) + torch.log2(size[:, 1] / scale)) / 2).int()
    gt_level = torch.min(torch.max(gt_level, lowest), highest)
    reach = ((points[None, :, :2] - centre[:, None, :]) / size[:, None, :]).norm(dim=2)            # [G, N]
    usable = gt_level[:, None] == level[None, :]
    if valid is not None:
        usable = usable & valid[None, :]
    inf = reach.new_full((), float('inf'))
    reach = torch.where(usable, reach, inf)
    if pos_scale_factor is None:
        near, which = torch.topk(reach, pos_num, dim=1, largest=False)
        claimed = torch.full_like(reach, float('inf')).scatter_(1, which, near)
    else:   # (the k = count(reach < factor) nearest are exactly the points within the factor)
        claimed = torch.where(reach < pos_scale_factor, reach, inf)
    best, owner = claimed.min(dim=0)                         # first minimum: the earliest ground truth keeps a tie
    return torch.where(best < inf, owner + 1, torch.zeros_like(owner))


def bbox_overlaps(bboxes1, bboxes2, mode='iou', is_aligned=False):
    """IoU / IoF (intersection over the FIRST box) with the +1 pixel convention: [rows, cols], or [rows] pairwise"""
    assert mode in ['iou', 'iof']
    rows, cols = bboxes1.size(0), bboxes2.size(0)
    if is_aligned:
        assert rows == cols
    if rows * cols == 0:
        return bboxes1.new(rows, 1) if is_aligned else bboxes1.new(rows, cols)
    first = bboxes1 if is_aligned else bboxes1[:, None, :]
    extent = (torch.min(first[..., 2:], bboxes2[..., 2:]) - torch.max(first[..., :2], bboxes2[..., :2]) + 1).clamp(min=0)
    shared = extent[..., 0] * extent[..., 1]
    area1 = (bboxes1[:, 2] - bboxes1[:, 0] + 1) * (bboxes1[:, 3] - bboxes1[:, 1] + 1)
    if not is_aligned:
        area1 = area1[:, None]
    if mode == 'iof':
        return shared / area1
    area2 = (bboxes2[:, 2] - bboxes2[:, 0] + 1) * (bboxes2[:, 3] - bboxes2[:, 1] + 1)
    return shared / (area1 + area2 - shared)


def assign_max_iou(overlaps, pos_iou_thr, neg_iou_thr, min_pos_iou=.0, gt_max_assign_all=True, valid=None):
    """MaxIoUAssigner's rule on a [G, N] overlap matrix -> ([N] long: -1 don't care, 0 negative, i > 0 ground truth
    i - 1; [N] best overlap).  A box is positive for its best ground truth from ``pos_iou_thr`` on, negative inside the
    ``neg_iou_thr`` range; then every ground truth whose best overlap reaches ``min_pos_iou`` takes its best box(es),
    the LAST such ground truth winning a box several of them claim.  Boxes with ``valid == False`` are left out of
    every maximum and come back as 0."""
    if valid is not None:
        overlaps = torch.where(valid[None, :], overlaps, overlaps.new_full((), -2.))
    num_gts = overlaps.shape[0]
    best, best_gt = overlaps.max(dim=0)
    top, top_box = overlaps.max(dim=1)
    low, high = (0, neg_iou_thr) if isinstance(neg_iou_thr, float) else (neg_iou_thr[0], neg_iou_thr[1])
    out = torch.full_like(best_gt, -1)
    out = torch.where((best >= low) & (best < high), torch.zeros_like(out), out)
    out = torch.where(best >= pos_iou_thr, best_gt + 1, out)
    keen = (top >= min_pos_iou)[:, None]
    if gt_max_assign_all:
        takes = (overlaps == top[:, None]) & keen
    else:
        takes = (torch.arange(overlaps.shape[1], device=overlaps.device)[None, :] == top_box[:, None]) & keen
    last = (takes.long() * torch.arange(1, num_gts + 1, device=overlaps.device)[:, None]).max(dim=0)[0]
    out = torch.where(last > 0, last, out)
    if valid is not None:
        out = torch.where(valid, out, torch.zeros_like(out))
    return out, best


# ------------------------------------------------------------------------------------------------
# the registry's assigner / sampler classes: front-ends
# ------------------------------------------------------------------------------------------------
class AssignResult(object):

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels

    def add_gt_(self, gt_labels):
        """prepend the ground truths themselves as samples (assigned to themselves, overlap 1)"""
        n = len(gt_labels)
        self.gt_inds = torch.cat([torch.arange(1, n + 1, dtype=torch.long, device=gt_labels.device), self.gt_inds])
        self.max_overlaps = torch.cat([self.max_overlaps.new_ones(self.num_gts), self.max_overlaps])
        if self.labels is not None:
            self.labels = torch.cat([gt_labels, self.labels])


class BaseAssigner(object):

    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None):
        raise NotImplementedError


def _need_both(candidates, gt_bboxes):
    if candidates.shape[0] == 0 or gt_bboxes.shape[0] == 0:
        raise ValueError('No gt or bboxes')


class PointAssigner(BaseAssigner):
    """0 = negative, i > 0 = positive for GT i-1 (``assign_points``)"""

    def __init__(self, scale=4, pos_num=3, pos_scale_factor=None):
        self.scale, self.pos_num, self.pos_scale_factor = scale, pos_num, pos_scale_factor

    def assign(self, points, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None, valid=None):
        _need_both(points, gt_bboxes)
        inds = assign_points(points, gt_bboxes, self.scale, self.pos_num, self.pos_scale_factor, valid)
        return AssignResult(gt_bboxes.shape[0], inds, None, labels=_gather_labels(inds, gt_labels))


class MaxIoUAssigner(BaseAssigner):
    """-1 = don't care, 0 = negative, i > 0 = positive for GT i-1 (``assign_max_iou``; the serial / parallel heads'
    refine stage)"""

    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=.0, gt_max_assign_all=True, ignore_iof_thr=-1,
                 ignore_wrt_candidates=True):
        self.pos_iou_thr, self.neg_iou_thr, self.min_pos_iou = pos_iou_thr, neg_iou_thr, min_pos_iou
        self.gt_max_assign_all, self.ignore_iof_thr, self.ignore_wrt_candidates = gt_max_assign_all, ignore_iof_thr, \
            ignore_wrt_candidates

    def _overlaps(self, bboxes, gt_bboxes, gt_bboxes_ignore):
        overlaps = bbox_overlaps(gt_bboxes, bboxes[:, :4])
        if self.ignore_iof_thr > 0 and gt_bboxes_ignore is not None and gt_bboxes_ignore.numel() > 0:
            # boxes (mostly) inside an ignore region are don't-care: every overlap of theirs becomes -1
            if self.ignore_wrt_candidates:
                covered = bbox_overlaps(bboxes[:, :4], gt_bboxes_ignore, mode='iof').max(dim=1)[0]
            else:
                covered = bbox_overlaps(gt_bboxes_ignore, bboxes[:, :4], mode='iof').max(dim=0)[0]
            overlaps = torch.where((covered > self.ignore_iof_thr)[None, :], overlaps.new_full((), -1.), overlaps)
        return overlaps

    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None, valid=None):
        _need_both(bboxes, gt_bboxes)
        return self.assign_wrt_overlaps(self._overlaps(bboxes, gt_bboxes, gt_bboxes_ignore), gt_labels, valid)

    def assign_wrt_overlaps(self, overlaps, gt_labels=None, valid=None):
        if overlaps.numel() == 0:
            raise ValueError('No gt or proposals')
        inds, best = assign_max_iou(overlaps, self.pos_iou_thr, self.neg_iou_thr, self.min_pos_iou,
                                    self.gt_max_assign_all, valid)
        return AssignResult(overlaps.size(0), inds, best, labels=_gather_labels(inds, gt_labels))


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
    """the positives / negatives of one image as index lists and gathered rows (reads the device: nonzero)"""

    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, gt_keypoints, assign_result, gt_flags):
        own = assign_result.gt_inds[pos_inds] - 1
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.pos_bboxes, self.neg_bboxes = bboxes[pos_inds], bboxes[neg_inds]
        self.pos_is_gt = gt_flags[pos_inds]
        self.num_gts = gt_bboxes.shape[0]
        self.pos_assigned_gt_inds = own
        self.pos_gt_bboxes, self.pos_gt_keypoints = gt_bboxes[own, :], gt_keypoints[own, :]
        self.pos_gt_labels = None if assign_result.labels is None else assign_result.labels[pos_inds]

    @property
    def bboxes(self):
        return torch.cat([self.pos_bboxes, self.neg_bboxes])


class PseudoSamplerKp(object):
    """no sampling: every positive / negative point is used"""

    def __init__(self, **kwargs):
        pass

    def sample(self, assign_result, bboxes, gt_bboxes, gt_keypoints, **kwargs):
        where = lambda m: torch.nonzero(m).reshape(-1)       # (ascending, unique)
        return SamplingResultKp(where(assign_result.gt_inds > 0), where(assign_result.gt_inds == 0), bboxes, gt_bboxes,
                                gt_keypoints, assign_result, bboxes.new_zeros(bboxes.shape[0], dtype=torch.uint8))


# ------------------------------------------------------------------------------------------------
# targets
# ------------------------------------------------------------------------------------------------
def unmap(data, count, inds, fill=0):
    """scatter the rows of a subset back into a tensor of ``count`` rows"""
    full = data.new_full((count, ) + tuple(data.shape[1:]), fill)
    full[inds] = data
    return full


def images_to_levels(target, num_level_grids):
    """[target_img0, target_img1] -> [target_level0, target_level1, ...]"""
    stacked = torch.stack(target, 0)
    pieces = torch.split(stacked, list(num_level_grids), dim=1)
    return [p.squeeze(0) for p in pieces]


def dense_targets_applicable(cfg, num_levels, all_valid=True, gt_bboxes_ignore_list=None):
    """the sync-free target path: any number of pyramid levels, any valid flags (round 6); PointAssigner with a fixed
    pos_num, or MaxIoUAssigner; no ignore regions"""
    a = cfg.assigner
    if not (gt_bboxes_ignore_list is None or all(g is None for g in gt_bboxes_ignore_list)):
        return False
    if a['type'] == 'PointAssigner':
        return a.get('pos_scale_factor') is None
    return a['type'] == 'MaxIoUAssigner'


def _assign_image(proposals, gt_bboxes, a, valid, gt_bboxes_ignore=None):
    """[N] assignment of one image's proposals under the assigner config ``a`` (a dict with ``type``)"""
    _need_both(proposals, gt_bboxes)
    if a['type'] == 'PointAssigner':
        return assign_points(proposals, gt_bboxes, a.get('scale', 4), a.get('pos_num', 3), a.get('pos_scale_factor'), valid)
    return build_assigner(dict(a)).assign(proposals, gt_bboxes, gt_bboxes_ignore, None, valid).gt_inds


def _image_targets(proposals, inds, gt_bboxes, gt_kps, gt_labels, pos_weight, valid):
    """the seven per-point target / weight tensors of one image from its assignment (zeros off the positives, label
    weight 1 on the negatives, 0 on don't-care and invalid points)"""
    pos = inds > 0
    own = (inds - 1).clamp(min=0)
    row, cell = pos[:, None], pos[:, None, None]
    kp = gt_kps[own]                                                    # [N, n_kp, 3]
    labels = torch.ones_like(inds) if gt_labels is None else gt_labels[own]
    counted = inds == 0 if valid is None else (inds == 0) & valid
    zero = proposals.new_zeros(())
    return (torch.where(pos, labels, torch.zeros_like(labels)),
            torch.where(pos, proposals.new_full((), pos_weight), counted.to(proposals.dtype)).contiguous(),
            torch.where(row, gt_bboxes[own], gt_bboxes.new_zeros(())),
            torch.where(row, proposals, zero),
            row.to(proposals.dtype).expand(-1, 4).contiguous(),
            torch.where(cell, kp[:, :, :2], kp.new_zeros(())),
            torch.where(cell, (kp[:, :, 2:3] != 0).to(proposals.dtype), kp.new_zeros(())).expand(-1, -1, 2).contiguous(),
            pos, counted)


def point_target_kp_dense(proposals_list, gt_bboxes_list, gt_kps_list, cfg, gt_labels_list=None, valid_flag_list=None):
    """The return value of ``point_target_kp(..., sampling=False)`` without a host read: per-level lists of labels, label
    weights, box targets, positive proposals, box weights, keypoint targets, keypoint weights, then num_total_pos /
    num_total_neg as 0-dim device tensors (sum over the images of max(count, 1)).  ``valid_flag_list``: per image a
    [N] bool tensor or a per-level list of them (None: every point valid)."""
    pos_weight = 1.0 if cfg.pos_weight <= 0 else cfg.pos_weight
    outs = [[] for _ in range(7)]
    total_pos = total_neg = None
    first = proposals_list[0]
    num_level = [p.shape[0] for p in first] if isinstance(first, (list, tuple)) else [first.shape[0]]
    for i, proposals in enumerate(proposals_list):
        proposals = torch.cat(proposals) if isinstance(proposals, (list, tuple)) else proposals
        valid = None if valid_flag_list is None else valid_flag_list[i]
        if isinstance(valid, (list, tuple)):
            valid = torch.cat(valid)
        if valid is not None:
            valid = valid.bool()
        labels = None if gt_labels_list is None else gt_labels_list[i]
        inds = _assign_image(proposals, gt_bboxes_list[i], cfg.assigner, valid)
        res = _image_targets(proposals, inds, gt_bboxes_list[i], gt_kps_list[i], labels, pos_weight, valid)
        for o, r in zip(outs, res[:7]):
            o.append(r)
        n_pos, n_neg = res[7].sum().clamp(min=1), res[8].sum().clamp(min=1)
        total_pos = n_pos if total_pos is None else total_pos + n_pos
        total_neg = n_neg if total_neg is None else total_neg + n_neg
    return tuple(images_to_levels(o, num_level) for o in outs) + (total_pos, total_neg)


def point_target_single(flat_proposals, valid_flags, gt_bboxes, gt_keypoints, gt_bboxes_ignore, gt_labels, cfg,
                        label_channels=1, sampling=True, unmap_outputs=True):
    """one image through the dense rules, returned the reference's way: seven tensors + the index lists of the positives
    and negatives among the VALID points (this front-end reads the device; the training path does not call it)"""
    valid = valid_flags.bool()
    if not valid.any():
        return (None, ) * 9
    if sampling:
        raise NotImplementedError('random samplers are not part of the KGDet path (focal loss => sampling=False)')
    a = cfg.assigner
    pos_weight = 1.0 if cfg.pos_weight <= 0 else cfg.pos_weight
    inds = _assign_image(flat_proposals, gt_bboxes, a, valid, gt_bboxes_ignore)
    res = _image_targets(flat_proposals, inds, gt_bboxes, gt_keypoints, gt_labels, pos_weight, valid)
    seven = res[:7] if unmap_outputs else tuple(r[valid] for r in res[:7])
    rank = torch.cumsum(valid.long(), 0) - 1                  # index of a valid point among the valid points
    return seven + (rank[res[7]], rank[res[8]])


def point_target_kp(proposals_list, valid_flag_list, gt_bboxes_list, gt_kps_list, img_metas, cfg,
                    gt_bboxes_ignore_list=None, gt_labels_list=None, label_channels=1, sampling=True,
                    unmap_outputs=True):
    """Targets of all images, regrouped per pyramid level; ``None`` if an image has no valid point.  num_total_pos /
    num_total_neg are host integers here (the reference's return type): use ``point_target_kp_dense`` in a step."""
    num_imgs = len(img_metas)
    assert len(proposals_list) == len(valid_flag_list) == num_imgs
    per_level = [pts.size(0) for pts in proposals_list[0]]
    flat, flags = [], []
    for pts, fl in zip(proposals_list, valid_flag_list):
        assert len(pts) == len(fl)
        flat.append(torch.cat(pts))
        flags.append(torch.cat(fl))
    ignore = gt_bboxes_ignore_list if gt_bboxes_ignore_list is not None else [None] * num_imgs
    labels = gt_labels_list if gt_labels_list is not None else [None] * num_imgs
    rows = [point_target_single(flat[i], flags[i], gt_bboxes_list[i], gt_kps_list[i], ignore[i], labels[i], cfg,
                                label_channels=label_channels, sampling=sampling, unmap_outputs=unmap_outputs)
            for i in range(num_imgs)]
    if any(r[0] is None for r in rows):
        return None
    counts = [sum(max(r[k].numel(), 1) for r in rows) for k in (7, 8)]
    return tuple(images_to_levels([r[k] for r in rows], per_level) for k in range(7)) + tuple(counts)
