"""Two-stage (init -> refine) keypoint RepPoints heads over the whole FPN pyramid:
``RepPointsHeadKpSerial`` and ``RepPointsHeadKpParallel`` (BASELINE config 5).

Host-side mirror of mmdet/models/anchor_heads/reppoints_head_kp_serial.py:16-750 (SER) and
reppoints_head_kp_parallel.py (identical except that the reppoints get their own conv /
deformable-conv branch instead of being regressed from the keypoints, :153-168, 314-315, 331-332).
Same registered names, constructor arguments and parameter names.  Per level one 3x3 deformable
conv for classification and one for keypoints (plus one for reppoints in the parallel head), all
through the HIP kernel; the init stage is assigned by ``PointAssigner``, the refine stage by
``MaxIoUAssigner`` on the boxes of the init reppoints (SER:551-573).

Reference quirks kept so that outputs stay identical: the decode step clamps ``kpts[:, 0::3]`` /
``kpts[:, 1::3]`` (the keypoint axis, SER:723-724) where the KGDet head clamps the coordinate axis;
keypoint weights are normalised per row without the x4 of the KGDet head (SER:469-477).
"""
from __future__ import division

import numpy as np
import os as _os

import torch
import torch.nn as nn

from . import conv1x1, dcn
from .heads import PointHeadMixin
from .layers import ConvModule, bias_init_with_prob, normal_init, shared_levels

GROUPED_REFINE = __import__('os').environ.get('KGDET_SERIAL_GROUPED_DCN', '1') == '1'   # 0: one call per deformable convolution (A/B)
SELECT_FIRST = __import__('os').environ.get('KGDET_SERIAL_SELECT_FIRST', '1') == '1'     # 0: convert whole maps, then keep nms_pre rows (A/B)
from .losses import SmoothL1Loss
import os

from .points import (PointGenerator, dense_targets_applicable, multi_apply, point_target_kp,
                     point_target_kp_dense)
from .postprocess import multiclass_nms_kp
from .registry import HEADS, build_loss

SHARED_LEVELS = os.environ.get('KGDET_SERIAL_SHARED_LEVELS', '1') == '1'     # 0: five autograd sub-graphs, engine-side gradient adds (A/B)
DENSE_TARGETS = os.environ.get('KGDET_SERIAL_DENSE_TARGETS', '1') == '1'     # 0: the reference-mirroring (host-syncing) path (A/B)


class _RepPointsHeadKpTwoStage(PointHeadMixin, nn.Module):
    parallel_reppts = False  # True: reppoints have their own conv branches (parallel head)

    def __init__(self,
                 num_classes,
                 in_channels,
                 feat_channels=256,
                 point_feat_channels=256,
                 stacked_convs=3,
                 num_reppts=9,
                 num_keypts=17,
                 gradient_mul=0.1,
                 point_strides=[8, 16, 32, 64, 128],
                 point_base_scale=4,
                 conv_cfg=None,
                 norm_cfg=None,
                 loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                 loss_bbox_init=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=0.5),
                 loss_bbox_refine=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0),
                 loss_kpt_init=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=0.5),
                 loss_kpt_refine=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0),
                 use_grid_points=False,
                 center_init=True,
                 transform_method='moment',
                 moment_mul=0.01):
        super().__init__()
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.feat_channels = feat_channels
        self.point_feat_channels = point_feat_channels
        self.stacked_convs = stacked_convs
        self.num_keypts = num_keypts
        self.num_reppts = num_reppts
        self.gradient_mul = gradient_mul
        self.point_base_scale = point_base_scale
        self.point_strides = point_strides
        self.conv_cfg = conv_cfg
        self.norm_cfg = norm_cfg
        self.use_sigmoid_cls = loss_cls.get('use_sigmoid', False)
        self.sampling = loss_cls['type'] not in ['FocalLoss']
        self.loss_cls = build_loss(loss_cls)
        self.loss_bbox_init = build_loss(loss_bbox_init)
        self.loss_bbox_refine = build_loss(loss_bbox_refine)
        self.loss_kpt_init = build_loss(loss_kpt_init)
        self.loss_kpt_refine = build_loss(loss_kpt_refine)
        self.use_grid_points = use_grid_points
        self.center_init = center_init
        self.transform_method = transform_method
        if self.transform_method == 'moment':
            self.moment_transfer = nn.Parameter(data=torch.zeros(2), requires_grad=True)
            self.moment_mul = moment_mul
        self.cls_out_channels = self.num_classes - 1 if self.use_sigmoid_cls else self.num_classes
        self.point_generators = [PointGenerator() for _ in self.point_strides]
        # the reppoints are the taps of a sqrt(n) x sqrt(n) deformable conv
        self.dcn_kernel = int(np.sqrt(num_reppts))
        self.dcn_pad = int((self.dcn_kernel - 1) / 2)
        assert self.dcn_kernel * self.dcn_kernel == num_reppts, 'The points number should be a square number.'
        assert self.dcn_kernel % 2 == 1, 'The points number should be an odd square number.'
        base = np.arange(-self.dcn_pad, self.dcn_pad + 1).astype(np.float64)
        yx = np.stack([np.repeat(base, self.dcn_kernel), np.tile(base, self.dcn_kernel)], axis=1).reshape(-1)
        self.dcn_base_offset = torch.tensor(yx).view(1, -1, 1, 1)
        self._init_layers()

    def _init_layers(self):
        self.relu = nn.ReLU(inplace=False)
        self.cls_convs = nn.ModuleList()
        self.reg_convs = nn.ModuleList()
        for i in range(self.stacked_convs):
            chn = self.in_channels if i == 0 else self.feat_channels
            self.cls_convs.append(ConvModule(chn, self.feat_channels, 3, stride=1, padding=1,
                                             conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg))
            self.reg_convs.append(ConvModule(chn, self.feat_channels, 3, stride=1, padding=1,
                                             conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg))
        keypts_out_dim = 2 * self.num_keypts
        reppts_out_dim = 2 * self.num_reppts
        fc, pc, k, pad = self.feat_channels, self.point_feat_channels, self.dcn_kernel, self.dcn_pad
        self.cls_refine_dfmconv = dcn.DeformConv(fc, pc, k, 1, pad)
        self.cls_refine_out = nn.Conv2d(pc, self.cls_out_channels, 1, 1, 0)
        self.keypts_init_conv = nn.Conv2d(fc, pc, 3, 1, 1)
        self.keypts_init_out = nn.Conv2d(pc, keypts_out_dim, 1, 1, 0)
        if self.parallel_reppts:
            self.reppts_init_conv = nn.Conv2d(fc, pc, 3, 1, 1)
        self.reppts_init_out = nn.Conv2d(pc if self.parallel_reppts else keypts_out_dim, reppts_out_dim, 1, 1, 0)
        self.keypts_refine_dfmconv = dcn.DeformConv(fc, pc, k, 1, pad)
        self.keypts_refine_out = nn.Conv2d(pc, keypts_out_dim, 1, 1, 0)
        if self.parallel_reppts:
            self.reppts_refine_dfmconv = dcn.DeformConv(fc, pc, k, 1, pad)
        self.reppts_refine_out = nn.Conv2d(pc if self.parallel_reppts else keypts_out_dim, reppts_out_dim, 1, 1, 0)

    def init_weights(self):
        for m in self.cls_convs:
            normal_init(m.conv, std=0.01)
        for m in self.reg_convs:
            normal_init(m.conv, std=0.01)
        bias_cls = bias_init_with_prob(0.01)
        normal_init(self.cls_refine_dfmconv, std=0.01)
        normal_init(self.cls_refine_out, std=0.01, bias=bias_cls)
        normal_init(self.keypts_init_conv, std=0.01)
        normal_init(self.keypts_init_out, std=0.01)
        normal_init(self.reppts_init_out, std=0.01)
        normal_init(self.keypts_refine_dfmconv, std=0.01)
        normal_init(self.keypts_refine_out, std=0.01)
        normal_init(self.reppts_refine_out, std=0.01)
        if self.parallel_reppts:
            normal_init(self.reppts_init_conv, std=0.01)
            normal_init(self.reppts_refine_dfmconv, std=0.01)

    # KGDET_SERIAL_CL=1 (A/B, off): inference under autocast with the towers on the pyramid's channels-last maps as they come --
    # MIOpen's NHWC kernels without the layout conversions it wraps around NCHW calls (~100 launches, ~1 ms of a batch of 8).
    # Measured 402.8 against 424.0 img/s: GroupNorm on a channels-last map reads 16 bytes of every 512 (a group's 8 channels of
    # a pixel) -- gn_split_moments 150 us against 51 us at 100 x 168 -- and the deformable stages need an NCHW copy per tower.
    channels_last_inference = _os.environ.get('KGDET_SERIAL_CL', '0') == '1'

    def _dfm(self, conv, feat, offset):
        """relu(deform_conv) with the ReLU fused into the kernel epilogue"""
        return dcn.deform_conv_cat(feat, [offset], [conv.weight], [self.dcn_pad])

    def forward_single(self, x):
        # (the reference keeps the regular grid as a plain CPU attribute and uploads it with type_as() on every call: a
        #  blocking host->device copy per level and step; uploaded once per device / dtype here)
        cache = self.__dict__.setdefault('_base_offset_cache', {})
        key = (x.device, x.dtype)
        if key not in cache:
            cache[key] = self.dcn_base_offset.to(device=x.device, dtype=x.dtype)
        dcn_base_offset = cache[key]
        if self.use_grid_points or not self.center_init:
            scale = self.point_base_scale / 2
            reppts_init = dcn_base_offset / dcn_base_offset.max() * scale
            keypts_init = 0
        else:
            reppts_init = 0
            keypts_init = 0
        cls_feat = x
        pts_feat = x
        for cls_conv in self.cls_convs:
            cls_feat = cls_conv(cls_feat)
        for reg_conv in self.reg_convs:
            pts_feat = reg_conv(pts_feat)
        # init stage
        # (the biased 3x3 convolutions + ReLU on the split-operand MFMA kernels, as in the KGDet head's first stage)
        keypts_out_init = conv1x1.conv_bias_act(self.keypts_init_out, conv1x1.conv_bias_act(self.keypts_init_conv, pts_feat, relu=True))
        if self.parallel_reppts:
            reppts_out_init = conv1x1.conv_bias_act(self.reppts_init_out, conv1x1.conv_bias_act(self.reppts_init_conv, pts_feat, relu=True))
        else:
            reppts_out_init = conv1x1.conv_bias_act(self.reppts_init_out, keypts_out_init)
        # (adding the integer 0 of the centre-init case would be a pass over a [B, 588, H, W] tensor for nothing)
        if not (isinstance(reppts_init, int) and reppts_init == 0):
            reppts_out_init = reppts_out_init + reppts_init
        if not (isinstance(keypts_init, int) and keypts_init == 0):
            keypts_out_init = keypts_out_init + keypts_init
        # refine stage: taps on the (gradient-scaled) init reppoints
        grad_mul = self.gradient_mul * reppts_out_init + (1 - self.gradient_mul) * reppts_out_init.detach()
        dcn_offset = grad_mul - dcn_base_offset
        # the refine stage's deformable convolutions share their offsets: ONE grouped call (one set of tap / inverse records,
        # grouped launches forward and backward, ReLU in the epilogue) instead of two or three
        if GROUPED_REFINE and cls_feat.is_cuda and not self.parallel_reppts:
            cls_dfm, kpt_dfm = dcn.deform_conv_cat_multi(
                [cls_feat, pts_feat], [dcn_offset],
                [[self.cls_refine_dfmconv.weight], [self.keypts_refine_dfmconv.weight]], [self.dcn_pad])
        else:
            cls_dfm = self._dfm(self.cls_refine_dfmconv, cls_feat, dcn_offset)
            kpt_dfm = self._dfm(self.keypts_refine_dfmconv, pts_feat, dcn_offset)
        cls_out = conv1x1.conv_bias_act(self.cls_refine_out, cls_dfm)
        keypts_out_refine = conv1x1.conv_bias_act(self.keypts_refine_out, kpt_dfm)
        if self.parallel_reppts:
            reppts_out_refine = conv1x1.conv_bias_act(self.reppts_refine_out, self._dfm(self.reppts_refine_dfmconv, pts_feat, dcn_offset))
        else:
            reppts_out_refine = conv1x1.conv_bias_act(self.reppts_refine_out, keypts_out_refine)
        keypts_out_refine = keypts_out_refine + keypts_out_init.detach()
        reppts_out_refine = reppts_out_refine + reppts_out_init.detach()
        return (cls_out, keypts_out_init, keypts_out_refine, reppts_out_init, reppts_out_refine)

    def forward(self, feats, img_metas):
        if SHARED_LEVELS and torch.is_grad_enabled() and feats[0].is_cuda and any(p.requires_grad for p in self.parameters()):
            # training: the five levels as ONE autograd node -- the shared parameters' gradients are summed over the levels by a few
            # multi-tensor launches inside it instead of 235 engine-side `add`s (layers.shared_levels)
            return shared_levels(self, self.forward_single, list(feats))
        return multi_apply(self.forward_single, feats)

    # ------------------------------------------------------------------------------------------
    def loss_single(self, cls_score, kpt_pred_init, kpt_pred_refine, rep_pred_init, rep_pred_refine, labels,
                    label_weights, bbox_gt_init, bbox_weights_init, bbox_gt_refine, bbox_weights_refine, kpt_gt_init,
                    kpt_weights_init, kpt_gt_refine, kpt_weights_refine, stride, num_total_samples_init,
                    num_total_samples_refine):
        labels = labels.reshape(-1)
        label_weights = label_weights.reshape(-1)
        cls_score = cls_score.permute(0, 2, 3, 1).reshape(-1, self.cls_out_channels)
        loss_cls = self.loss_cls(cls_score, labels, label_weights, avg_factor=num_total_samples_refine)

        normalize_term = self.point_base_scale * stride
        bbox_pred_init = self.points2bbox(rep_pred_init.reshape(-1, 2 * self.num_reppts), y_first=False)
        bbox_pred_refine = self.points2bbox(rep_pred_refine.reshape(-1, 2 * self.num_reppts), y_first=False)
        # (SmoothL1Loss takes the normalisation as `divisor`: the fused op reads the raw tensors -- no pred / d, target / d
        #  passes over [points, 588] tensors; same values)
        def normalised(loss_fn, pred, gt, weights, avg):
            if isinstance(loss_fn, SmoothL1Loss):
                return loss_fn(pred, gt, weights, avg_factor=avg, divisor=normalize_term)
            return loss_fn(pred / normalize_term, gt / normalize_term, weights, avg_factor=avg)

        loss_bbox_init = normalised(self.loss_bbox_init, bbox_pred_init, bbox_gt_init.reshape(-1, 4),
                                    bbox_weights_init.reshape(-1, 4), num_total_samples_init)
        loss_bbox_refine = normalised(self.loss_bbox_refine, bbox_pred_refine, bbox_gt_refine.reshape(-1, 4),
                                      bbox_weights_refine.reshape(-1, 4), num_total_samples_refine)

        def kpt_loss(loss_fn, pred, gt, weights, avg):
            # SER:469-477 normalises in place; with one image per GPU the per-level targets are views of one
            # tensor and the in-place update would invalidate weights autograd saved for another level, so
            # the normalisation runs on a private copy (same values)
            # (rows without a visible keypoint are all zero: dividing them by 1 gives the same tensor as the reference's
            #  boolean-mask update, without its device->host round trip)
            weights = weights.reshape(-1, self.num_keypts * 2)
            weights = weights / weights.sum(1).clamp(min=1).unsqueeze(1)
            return normalised(loss_fn, pred.reshape(-1, self.num_keypts * 2), gt.reshape(-1, self.num_keypts * 2), weights, avg)

        loss_kpt_init = kpt_loss(self.loss_kpt_init, kpt_pred_init, kpt_gt_init, kpt_weights_init,
                                 num_total_samples_init)
        loss_kpt_refine = kpt_loss(self.loss_kpt_refine, kpt_pred_refine, kpt_gt_refine, kpt_weights_refine,
                                   num_total_samples_refine)
        return loss_cls, loss_bbox_init, loss_bbox_refine, loss_kpt_init, loss_kpt_refine

    def loss(self, cls_scores, keypts_preds_init, keypts_preds_refine, reppts_preds_init, reppts_preds_refine,
             gt_bboxes, gt_labels, gt_keypoints, img_metas, cfg, gt_bboxes_ignore=None):
        featmap_sizes = [featmap.size()[-2:] for featmap in cls_scores]
        assert len(featmap_sizes) == len(self.point_generators)
        label_channels = self.cls_out_channels if self.use_sigmoid_cls else 1
        device = cls_scores[0].device

        # init stage targets
        center_list, valid_flag_list = self.get_points(featmap_sizes, img_metas, device=device)
        kpt_coord_init = self.offset_to_pts(center_list, keypts_preds_init)
        rep_coord_init = self.offset_to_pts(center_list, reppts_preds_init)
        if cfg.init.assigner['type'] == 'PointAssigner':
            candidate_list = center_list
        else:
            candidate_list = self.centers_to_bboxes(center_list)
        # every grid point valid?  (host arithmetic on the image metas; if not, the dense path takes the valid flags)
        all_valid = all(
            min(int(np.ceil(meta['pad_shape'][0] / s)), fs[0]) == fs[0] and
            min(int(np.ceil(meta['pad_shape'][1] / s)), fs[1]) == fs[1]
            for meta in img_metas for s, fs in zip(self.point_strides, featmap_sizes))
        dense_ok = lambda c: (not self.sampling and DENSE_TARGETS and
                              dense_targets_applicable(c, len(self.point_strides), all_valid, gt_bboxes_ignore))
        if dense_ok(cfg.init):
            targets_init = point_target_kp_dense(candidate_list, gt_bboxes, gt_keypoints, cfg.init, gt_labels_list=gt_labels,
                                                 valid_flag_list=None if all_valid else valid_flag_list)
        else:
            targets_init = point_target_kp(candidate_list, valid_flag_list, gt_bboxes, gt_keypoints, img_metas, cfg.init,
                                           gt_bboxes_ignore_list=gt_bboxes_ignore, gt_labels_list=gt_labels,
                                           label_channels=label_channels, sampling=self.sampling)
        (*_, bbox_gt_list_init, candidate_list_init, bbox_weights_list_init, keypoint_gt_list_init,
         keypoint_weights_list_init, num_total_pos_init, num_total_neg_init) = targets_init
        num_total_samples_init = (num_total_pos_init + num_total_neg_init if self.sampling else num_total_pos_init)

        # refine stage targets: boxes of the (detached) init reppoints, MaxIoUAssigner
        center_list, valid_flag_list = self.get_points(featmap_sizes, img_metas, device=device)
        kpt_coord_refine = self.offset_to_pts(center_list, keypts_preds_refine)
        rep_coord_refine = self.offset_to_pts(center_list, reppts_preds_refine)
        init_boxes = [self.points2bbox(reppts_preds_init[i_lvl].detach()) * self.point_strides[i_lvl]
                      for i_lvl in range(len(reppts_preds_refine))]
        bbox_list = []
        for i_img, center in enumerate(center_list):
            bbox = []
            for i_lvl in range(len(reppts_preds_refine)):
                bbox_center = torch.cat([center[i_lvl][:, :2], center[i_lvl][:, :2]], dim=1)
                bbox.append(bbox_center + init_boxes[i_lvl][i_img].permute(1, 2, 0).reshape(-1, 4))
            bbox_list.append(bbox)
        if dense_ok(cfg.refine):
            targets_refine = point_target_kp_dense(bbox_list, gt_bboxes, gt_keypoints, cfg.refine, gt_labels_list=gt_labels,
                                                   valid_flag_list=None if all_valid else valid_flag_list)
        else:
            targets_refine = point_target_kp(bbox_list, valid_flag_list, gt_bboxes, gt_keypoints, img_metas, cfg.refine,
                                             gt_bboxes_ignore_list=gt_bboxes_ignore, gt_labels_list=gt_labels,
                                             label_channels=label_channels, sampling=self.sampling)
        (labels_list, label_weights_list, bbox_gt_list_refine, candidate_list_refine, bbox_weights_list_refine,
         keypoint_gt_list_refine, keypoint_weights_list_refine, num_total_pos_refine,
         num_total_neg_refine) = targets_refine
        num_total_samples_refine = (num_total_pos_refine + num_total_neg_refine
                                    if self.sampling else num_total_pos_refine)

        per_level = multi_apply(self.loss_single, cls_scores, kpt_coord_init, kpt_coord_refine, rep_coord_init,
                                rep_coord_refine, labels_list, label_weights_list, bbox_gt_list_init,
                                bbox_weights_list_init, bbox_gt_list_refine, bbox_weights_list_refine,
                                keypoint_gt_list_init, keypoint_weights_list_init, keypoint_gt_list_refine,
                                keypoint_weights_list_refine, self.point_strides,
                                num_total_samples_init=num_total_samples_init,
                                num_total_samples_refine=num_total_samples_refine)
        names = ['loss_cls', 'loss_bbox_init', 'loss_bbox_refine', 'loss_kpt_init', 'loss_kpt_refine']
        return dict(zip(names, per_level))

    # ------------------------------------------------------------------------------------------
    def get_bboxes(self, cls_scores, keypts_preds_init, keypts_preds_refine, reppts_preds_init, reppts_preds_refine,
                   img_metas, cfg, rescale=False, nms=True):
        assert len(cls_scores) == len(keypts_preds_refine) == len(reppts_preds_refine)
        cls_scores = [t.float() for t in cls_scores]             # decode in fp32 under autocast
        bbox_preds_refine = [self.points2bbox(r.float()) for r in reppts_preds_refine]
        kpt_preds_refine = [self.points2kpt(k.float()) for k in keypts_preds_refine]
        num_levels = len(cls_scores)
        device = cls_scores[0].device
        mlvl_points = [self.point_generators[i].grid_points(cls_scores[i].size()[-2:], self.point_strides[i],
                                                            device=device) for i in range(num_levels)]
        result_list = []
        for img_id in range(len(img_metas)):
            result_list.append(self.get_bboxes_single(
                [cls_scores[i][img_id].detach() for i in range(num_levels)],
                [bbox_preds_refine[i][img_id].detach() for i in range(num_levels)],
                [kpt_preds_refine[i][img_id].detach() for i in range(num_levels)], mlvl_points,
                img_metas[img_id]['img_shape'], img_metas[img_id]['scale_factor'], cfg, rescale, nms))
        return result_list

    def get_bboxes_single(self, cls_scores, bbox_preds, kpt_preds, mlvl_points, img_shape, scale_factor, cfg,
                          rescale=False, nms=True):
        assert len(cls_scores) == len(bbox_preds) == len(mlvl_points) == len(kpt_preds)
        mlvl_bboxes, mlvl_kpts, mlvl_scores = [], [], []
        num_kpt = self.num_keypts
        num_kp_channel = kpt_preds[0].size(0) // num_kpt
        assert num_kp_channel == 2 or num_kp_channel == 3
        for i_lvl, (cls_score, bbox_pred, kpt_pred, points) in enumerate(zip(cls_scores, bbox_preds, kpt_preds,
                                                                             mlvl_points)):
            assert cls_score.size()[-2:] == bbox_pred.size()[-2:] == kpt_pred.size()[-2:]
            cls_score = cls_score.permute(1, 2, 0).reshape(-1, self.cls_out_channels)
            scores = cls_score.sigmoid() if self.use_sigmoid_cls else cls_score.softmax(-1)
            bbox_pred = bbox_pred.permute(1, 2, 0).reshape(-1, 4)
            if num_kp_channel == 3:
                kpt_pred = kpt_pred.permute(1, 2, 0).reshape(-1, num_kpt * num_kp_channel)
            else:
                kpt_pred = kpt_pred.permute(1, 2, 0).reshape(-1, num_kpt, num_kp_channel)
                kpt_pred = torch.cat([kpt_pred, kpt_pred.new_full(kpt_pred[:, :, :1].size(), 1)], dim=2)
                kpt_pred = kpt_pred.reshape(-1, num_kpt * 3)
            nms_pre = cfg.get('nms_pre', -1)
            if nms_pre > 0 and scores.shape[0] > nms_pre:
                max_scores, _ = scores.max(dim=1) if self.use_sigmoid_cls else scores[:, 1:].max(dim=1)
                _, topk_inds = max_scores.topk(nms_pre)
                points = points[topk_inds, :]
                bbox_pred = bbox_pred[topk_inds, :]
                kpt_pred = kpt_pred[topk_inds, :]
                scores = scores[topk_inds, :]
            bbox_pos_center = torch.cat([points[:, :2], points[:, :2]], dim=1)
            bboxes = bbox_pred * self.point_strides[i_lvl] + bbox_pos_center
            kpts = kpt_pred.view(-1, num_kpt, 3).clone()
            kpts[:, :, :2] = kpts[:, :, :2] * self.point_strides[i_lvl] + points[:, :2].unsqueeze(dim=1)
            bboxes = torch.stack([bboxes[:, 0].clamp(min=0, max=img_shape[1]),
                                  bboxes[:, 1].clamp(min=0, max=img_shape[0]),
                                  bboxes[:, 2].clamp(min=0, max=img_shape[1]),
                                  bboxes[:, 3].clamp(min=0, max=img_shape[0])], dim=-1)
            # reference quirk (SER:723-724): the slice runs over the keypoint axis, not the coordinate axis
            kpts[:, 0::3] = kpts[:, 0::3].clamp(min=0, max=img_shape[1])
            kpts[:, 1::3] = kpts[:, 1::3].clamp(min=0, max=img_shape[0])
            mlvl_bboxes.append(bboxes)
            mlvl_scores.append(scores)
            mlvl_kpts.append(kpts)
        mlvl_bboxes = torch.cat(mlvl_bboxes)
        mlvl_kpts = torch.cat(mlvl_kpts)
        if rescale:
            mlvl_bboxes /= mlvl_bboxes.new_tensor(scale_factor)
            mlvl_kpts[:, :, 0:2] = mlvl_kpts[:, :, 0:2] / mlvl_kpts.new_tensor(scale_factor)
            mlvl_kpts = mlvl_kpts.reshape(-1, num_kpt * 3)
        mlvl_scores = torch.cat(mlvl_scores)
        if self.use_sigmoid_cls:
            mlvl_scores = torch.cat([mlvl_scores.new_zeros(mlvl_scores.shape[0], 1), mlvl_scores], dim=1)
        if nms:
            return multiclass_nms_kp(mlvl_bboxes, mlvl_scores, mlvl_kpts, cfg.score_thr, cfg.nms, cfg.max_per_img)
        return mlvl_bboxes, mlvl_scores, mlvl_kpts


    # ------------------------------------------------------------------------------------------
    # whole-batch decode + fused (soft-)NMS: no per-image / per-class Python loop, no host read before the
    # results -- capturable as one HIP graph (detector.graphed_test_batch).  Same values as get_bboxes_single.
    # ------------------------------------------------------------------------------------------
    def _packed_ok(self, cls_scores, img_metas, cfg, rescale):
        kind = cfg.nms.get('type', 'nms')
        if not (cls_scores[0].is_cuda and kind in ('nms', 'soft_nms') and self.use_sigmoid_cls and cfg.max_per_img > 0
                and rescale):      # (rescale=False leaves the landmarks [n, K, 3]: SER:730-734 reshapes only when rescaling)
            return False
        if not all(isinstance(m['scale_factor'], (int, float)) for m in img_metas):
            return False
        pre = cfg.get('nms_pre', -1)
        n = sum(min(c.shape[-2] * c.shape[-1], pre) if pre > 0 else c.shape[-2] * c.shape[-1] for c in cls_scores)
        C = self.cls_out_channels
        if kind == 'nms':
            return n <= 4096 and n * C <= 16384 and C <= 64
        if cfg.nms.get('method', 'linear') not in ('linear', 'gaussian'):
            return False
        from .postprocess import soft_nms_fused_supported
        return soft_nms_fused_supported(len(img_metas), n, C, cfg.max_per_img)

    def _decode_level_batch(self, cls_score, bbox_pred, kpt_pred, points, stride, lim_w, lim_h, cfg, selected=None):
        """one level of get_bboxes_single for all images at once: boxes [B,n,4], scores [B,n,C], landmarks [B,n,K,3].
        selected = (scores [B,k,C], centres [B,k,2]): the nms_pre candidates were chosen beforehand and bbox_pred / kpt_pred hold
        THEIR rows as [B, 4, k, 1] / [B, K ch, k, 1] (get_bboxes_packed_tensor: the conversions then run on k = 1000 rows per
        level instead of the whole map -- 474 MB of landmark rows at 100 x 168 for a batch of 8)"""
        B, num_kpt = cls_score.shape[0] if selected is None else selected[0].shape[0], self.num_keypts
        ch = kpt_pred.size(1) // num_kpt
        assert ch == 2 or ch == 3
        if selected is None:
            scores = cls_score.permute(0, 2, 3, 1).reshape(B, -1, self.cls_out_channels).sigmoid()
        else:
            scores = selected[0]
        bbox_pred = bbox_pred.permute(0, 2, 3, 1).reshape(B, -1, 4)
        kpt_pred = kpt_pred.permute(0, 2, 3, 1).reshape(B, -1, num_kpt, ch)
        if ch == 2:
            kpt_pred = torch.cat([kpt_pred, kpt_pred.new_ones(kpt_pred[..., :1].shape)], dim=-1)
        ctr = points[:, :2].unsqueeze(0).expand(B, -1, -1) if selected is None else selected[1]
        nms_pre = cfg.get('nms_pre', -1)
        if selected is None and nms_pre > 0 and scores.shape[1] > nms_pre:
            _, top = scores.max(dim=2)[0].topk(nms_pre, dim=1)
            ctr = torch.gather(ctr, 1, top.unsqueeze(-1).expand(B, nms_pre, 2))
            bbox_pred = torch.gather(bbox_pred, 1, top.unsqueeze(-1).expand(B, nms_pre, 4))
            kpt_pred = torch.gather(kpt_pred, 1, top.view(B, nms_pre, 1, 1).expand(B, nms_pre, num_kpt, 3))
            scores = torch.gather(scores, 1, top.unsqueeze(-1).expand(B, nms_pre, scores.shape[2]))
        bboxes = bbox_pred * stride + torch.cat([ctr, ctr], dim=2)
        kpts = kpt_pred.clone()
        kpts[..., :2] = kpts[..., :2] * stride + ctr.unsqueeze(2)

        def clamp(t, lim):
            if isinstance(lim, (int, float)):
                return t.clamp(min=0, max=lim)
            return torch.minimum(t.clamp(min=0), lim.view((B, ) + (1, ) * (t.dim() - 1)))

        bboxes = torch.stack([clamp(bboxes[..., 0], lim_w), clamp(bboxes[..., 1], lim_h), clamp(bboxes[..., 2], lim_w),
                              clamp(bboxes[..., 3], lim_h)], dim=-1)
        # reference quirk kept (SER:723-724): the slices run over the KEYPOINT axis -- every third landmark's (x, y, v)
        kpts[:, :, 0::3] = clamp(kpts[:, :, 0::3], lim_w)
        kpts[:, :, 1::3] = clamp(kpts[:, :, 1::3], lim_h)
        return bboxes, scores, kpts

    def get_bboxes_packed(self, cls_scores, bbox_preds, kpt_preds, mlvl_points, img_metas, cfg, rescale=True, selected=None):
        """refine-stage maps -> fixed-size device tensors (det [B,M,5], labels [B,M], landmarks [B,M,3K], count [B])"""
        from .heads import RepPointsHeadKp3RepCas1AssignOnce as _K
        from .postprocess import multiclass_nms_kp_fused, multiclass_soft_nms_kp_fused
        device = cls_scores[0].device
        lim_w = _K._per_image([float(m['img_shape'][1]) for m in img_metas], device)
        lim_h = _K._per_image([float(m['img_shape'][0]) for m in img_metas], device)
        decoded = [self._decode_level_batch(cls_scores[i].detach().float(), bbox_preds[i].detach().float(),
                                            kpt_preds[i].detach().float(), mlvl_points[i], self.point_strides[i],
                                            lim_w, lim_h, cfg, None if selected is None else selected[i])
                   for i in range(len(cls_scores))]
        bboxes = torch.cat([d[0] for d in decoded], dim=1)
        scores = torch.cat([d[1] for d in decoded], dim=1)
        kpts = torch.cat([d[2] for d in decoded], dim=1)
        # get_bboxes_single divides by `new_tensor(scale_factor)`: a TRUE division (by a python scalar torch multiplies with
        # the reciprocal -- one ulp off for factors like 1.5), so the divisor is a tensor here as well; a fill kernel when the
        # images agree (capturable in a graph), one non-blocking upload otherwise
        sf = _K._per_image([float(m['scale_factor']) for m in img_metas], device)
        if isinstance(sf, float):
            sf = torch.full((bboxes.shape[0], 1), sf, dtype=torch.float32, device=device)
        bboxes = bboxes / sf.view(-1, 1, 1)
        kpts[..., 0:2] = kpts[..., 0:2] / sf.view(-1, 1, 1, 1)
        kpts = kpts.reshape(kpts.shape[0], kpts.shape[1], -1)
        if cfg.nms.get('type', 'nms') == 'soft_nms':
            return multiclass_soft_nms_kp_fused(bboxes, scores, kpts, cfg.score_thr, cfg.nms, cfg.max_per_img)
        return multiclass_nms_kp_fused(bboxes, scores, kpts, cfg.score_thr, float(cfg.nms['iou_thr']), cfg.max_per_img)

    def get_bboxes_packed_tensor(self, cls_scores, keypts_preds_init, keypts_preds_refine, reppts_preds_init,
                                 reppts_preds_refine, img_metas, cfg, rescale=False):
        """The batch's detections as ONE device tensor [B, max_per_img, 7 + 3K] -- box (4), score, label, count (repeated),
        landmarks -- produced without any host read; None when the packed path does not apply."""
        cls = [t.float() for t in cls_scores]
        if not self._packed_ok(cls, img_metas, cfg, rescale):
            return None
        points = [self.point_generators[i].grid_points(cls[i].size()[-2:], self.point_strides[i], device=cls[i].device)
                  for i in range(len(cls))]
        nms_pre = cfg.get('nms_pre', -1)
        if SELECT_FIRST and nms_pre > 0:
            # The nms_pre candidates of a level are a function of its scores alone: choose them FIRST and convert only their
            # rows of the 588-channel point maps (get_bboxes_single, like the reference, converts every location of the map and
            # then keeps 1000: for a batch of 8 at 100 x 168 the float cast, the (y, x) -> (x, y) swap and the visibility column
            # moved ~1.5 GB).  Gathering rows commutes with the per-location arithmetic: same values.
            box, kpt, selected = [], [], []
            for i in range(len(cls)):
                B, n = cls[i].shape[0], cls[i].shape[2] * cls[i].shape[3]
                scores = cls[i].permute(0, 2, 3, 1).reshape(B, n, self.cls_out_channels).sigmoid()
                ctr = points[i][:, :2].unsqueeze(0).expand(B, -1, -1)
                rep, kp_ = reppts_preds_refine[i].detach().reshape(B, -1, n), keypts_preds_refine[i].detach().reshape(B, -1, n)
                if n > nms_pre:
                    _, top = scores.max(dim=2)[0].topk(nms_pre, dim=1)
                    ctr = torch.gather(ctr, 1, top.unsqueeze(-1).expand(B, nms_pre, 2))
                    scores = torch.gather(scores, 1, top.unsqueeze(-1).expand(B, nms_pre, scores.shape[2]))
                    rep = torch.gather(rep, 2, top.unsqueeze(1).expand(B, rep.shape[1], nms_pre))
                    kp_ = torch.gather(kp_, 2, top.unsqueeze(1).expand(B, kp_.shape[1], nms_pre))
                box.append(self.points2bbox(rep.float().unsqueeze(-1)))
                kpt.append(self.points2kpt(kp_.float().unsqueeze(-1)))
                selected.append((scores, ctr))
            det, label, kp, count = self.get_bboxes_packed(cls, box, kpt, points, img_metas, cfg, rescale, selected=selected)
        else:
            box = [self.points2bbox(r.float()) for r in reppts_preds_refine]
            kpt = [self.points2kpt(k.float()) for k in keypts_preds_refine]
            det, label, kp, count = self.get_bboxes_packed(cls, box, kpt, points, img_metas, cfg, rescale)
        B, M = label.shape
        return torch.cat([det, label.unsqueeze(-1).float(), count.view(B, 1, 1).expand(B, M, 1).float(), kp], dim=-1)

    @staticmethod
    def unpack_results(packed):
        """host copy of ``get_bboxes_packed_tensor`` -> per image (det [n, 5], labels [n] int64, landmarks [n, 3K])"""
        out = []
        for b in range(packed.shape[0]):
            n = int(packed[b, 0, 6])
            out.append((packed[b, :n, :5], packed[b, :n, 5].astype(np.int64), packed[b, :n, 7:]))
        return out

    def get_bboxes_numpy(self, *args, **kwargs):
        """``get_bboxes`` with the results on the host as numpy arrays; on the packed path ONE device->host copy per batch"""
        packed = self.get_bboxes_packed_tensor(*args, **kwargs)
        if packed is None:
            return [(d.float().cpu().numpy(), lab.cpu().numpy(), k.float().cpu().numpy().reshape(k.shape[0], -1))
                    for d, lab, k in self.get_bboxes(*args, **kwargs)]
        return self.unpack_results(packed.cpu().numpy())


@HEADS.register_module
class RepPointsHeadKpSerial(_RepPointsHeadKpTwoStage):
    """reppoints regressed from the keypoint maps (1x1 conv on keypts_out)"""
    parallel_reppts = False


@HEADS.register_module
class RepPointsHeadKpParallel(_RepPointsHeadKpTwoStage):
    """reppoints predicted by their own conv / deformable-conv branch"""
    parallel_reppts = True
