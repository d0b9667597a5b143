"""The KGDet head: ``Kp3RepBlock`` and ``RepPointsHeadKp3RepCas1AssignOnce``.

Host-side mirror of mmdet/models/anchor_heads/reppoints_head_kp3rep_cas_1_assign_once.py (KP3):
same registered name, constructor arguments, sub-module / parameter names (checkpoint-key
contract: ``cls_convs.{i}.conv|gn``, ``kp_rep_block_{1,2,3}.{cls,keypts}_dfmconv_{3,5,7}.weight``,
``moment_transfer`` ...), same outputs.  What differs is how the work reaches the GPU:

* the three deformable convs of a branch (3x3 / 5x5 / 7x7, KP3:145-153) run through
  ``dcn.deform_conv_cat``: ReLU fused into the kernel epilogue and each conv writing its channel
  window of one [B,768,H,W] buffer (no cat, no ReLU launches);
* ``points2bbox`` ('moment', KP3:373-388) is one fused HIP kernel each way (``moment.moment_bbox``);
* decoding + NMS of a whole batch goes through one batched NMS launch
  (``postprocess.multiclass_nms_kp_batched``); the reference handles one image, one class at a time
  with a device->host sync per class.
``KGDetHead`` is registered as an alias of the class, never a replacement.
"""
from __future__ import division

import ctypes
import os

import numpy as np
import torch
import torch.nn as nn

from . import conv1x1, dcn, moment
from .layers import ConvModule, bias_init_with_prob, normal_init
from .points import (PointGenerator, dense_targets_applicable, multi_apply, point_target_kp,
                     point_target_kp_dense)
from .postprocess import multiclass_nms_kp, multiclass_nms_kp_batched
from .registry import HEADS, build_loss

_KERNELS = (3, 5, 7)           # the three deformable kernel sizes of a Kp3RepBlock
_GROUP_POINTS = (9, 25, 49)    # reppoints consumed by each (KP3:131-133)


def _base_offset(k):
    """regular k x k sampling grid as (y, x) pairs, row-major, [1, 2*k*k, 1, 1] (KP3:37-46)"""
    pad = (k - 1) // 2
    base = np.arange(-pad, pad + 1).astype(np.float64)
    yx = np.stack([np.repeat(base, k), np.tile(base, k)], axis=1).reshape(-1)
    return torch.tensor(yx).view(1, -1, 1, 1)


_FUSED_OFFSETS = os.environ.get('KGDET_FUSED_OFFSETS', '1') == '1'     # 0: the torch chain (A/B)


class _PtsFromOffsets(torch.autograd.Function):
    """offsets [B, 2n, H, W] -> image coordinates [B, H*W, 2n] = offset * stride + centre, (x, y) interleaved: the permute /
    flip / multiply / add chain of offset_to_pts (SER:400-421) as one pass each way (csrc/glue.hip); same values bit for bit"""

    @staticmethod
    def forward(ctx, pred, centres, stride, y_first):
        from . import _lib
        pred = pred.contiguous()
        B, C, H, W = pred.shape
        pts = pred.new_empty(B, H * W, C)
        _lib.check(_lib.lib().kgdet_pts_from_offsets_forward(
            _lib.ptr(pred), _lib.ptr(centres.contiguous()), _lib.ptr(pts), ctypes.c_int64(B), ctypes.c_int32(C),
            ctypes.c_int64(H * W), ctypes.c_float(stride), ctypes.c_int32(1 if y_first else 0), _lib.current_stream()),
            'pts_from_offsets_forward')
        ctx.shape, ctx.stride, ctx.y_first = (B, C, H, W), stride, y_first
        return pts

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        from . import _lib
        B, C, H, W = ctx.shape
        g = g.contiguous()
        grad = g.new_empty(B, C, H, W)
        _lib.check(_lib.lib().kgdet_pts_from_offsets_backward(
            _lib.ptr(g), _lib.ptr(grad), ctypes.c_int64(B), ctypes.c_int32(C), ctypes.c_int64(H * W),
            ctypes.c_float(ctx.stride), ctypes.c_int32(1 if ctx.y_first else 0), _lib.current_stream()),
            'pts_from_offsets_backward')
        return grad, None, None, None


class _RepOffsets(torch.autograd.Function):
    """reppoints [B, >= 166, H, W] -> the three offset tensors of a Kp3RepBlock; the value is the reference's
    ``gm * part + (1 - gm) * part.detach() - base`` expression, the gradient ``gm * grad`` (csrc/glue.hip)"""

    @staticmethod
    def forward(ctx, reppts, gm):
        from . import _lib
        reppts = reppts.contiguous()
        B, C, H, W = reppts.shape
        ks = (ctypes.c_int32 * 3)(*_KERNELS)
        outs = [reppts.new_empty(B, 2 * k * k, H, W) for k in _KERNELS]
        _lib.check(_lib.lib().kgdet_reppts_offsets_forward(
            _lib.ptr(reppts), ctypes.c_int32(B), ctypes.c_int32(C), ctypes.c_int32(H * W), ks, ctypes.c_float(gm),
            _lib.ptr(outs[0]), _lib.ptr(outs[1]), _lib.ptr(outs[2]), _lib.current_stream()), 'reppts_offsets_forward')
        ctx.gm, ctx.shape = gm, (B, C, H, W)
        return tuple(outs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g0, g1, g2):
        from . import _lib
        B, C, H, W = ctx.shape
        gs = [None if g is None else g.contiguous() for g in (g0, g1, g2)]
        like = next(g for g in gs if g is not None)
        grad = like.new_empty(B, C, H, W)
        ks = (ctypes.c_int32 * 3)(*_KERNELS)
        _lib.check(_lib.lib().kgdet_reppts_offsets_backward(
            _lib.ptr(gs[0]), _lib.ptr(gs[1]), _lib.ptr(gs[2]), ctypes.c_int32(B), ctypes.c_int32(C), ctypes.c_int32(H * W), ks,
            ctypes.c_float(ctx.gm), _lib.ptr(grad), _lib.current_stream()), 'reppts_offsets_backward')
        return grad, None


class Kp3RepBlock(nn.Module):
    """One cascade stage: classification / keypoint / reppoint maps from the two tower features,
    through plain 3x3 convs (``deform_conv=False``) or through three deformable convs whose taps
    sit on the previous stage's 9+25+49 reppoints (``deform_conv=True``)."""

    def __init__(self, deform_conv, cls_out_channels, in_channels=256, feat_channels=256, num_reppts=9,
                 num_keypts=17, gradient_mul=0.1):
        super().__init__()
        self.deform_conv = deform_conv
        self.gradient_mul = gradient_mul
        keypts_out_dim = 2 * num_keypts
        reppts_out_dim = 2 * num_reppts
        self.relu = nn.ReLU(inplace=False)

        if deform_conv:
            for k in _KERNELS:
                setattr(self, 'dcn_kernel_%d' % k, k)
                setattr(self, 'dcn_pad_%d' % k, (k - 1) // 2)
                setattr(self, 'dcn_base_offset_%d' % k, _base_offset(k))  # plain attribute, not a buffer
                setattr(self, 'cls_dfmconv_%d' % k, dcn.DeformConv(in_channels, feat_channels, k, 1, (k - 1) // 2))
            self.cls_out = nn.Conv2d(feat_channels * 3, cls_out_channels, 1, 1, 0)
            for k in _KERNELS:
                setattr(self, 'keypts_dfmconv_%d' % k,
                        dcn.DeformConv(in_channels, feat_channels, k, 1, (k - 1) // 2))
            self.keypts_out = nn.Conv2d(feat_channels * 3, keypts_out_dim, 1, 1, 0)
            self.reppts_out = nn.Conv2d(keypts_out_dim, reppts_out_dim, 1, 1, 0)
        else:
            self.cls_conv = nn.Conv2d(in_channels, feat_channels, 3, 1, 1)
            self.cls_out = nn.Conv2d(feat_channels, cls_out_channels, 1, 1, 0)
            self.keypts_conv = nn.Conv2d(in_channels, feat_channels, 3, 1, 1)
            self.keypts_out = nn.Conv2d(feat_channels, keypts_out_dim, 1, 1, 0)
            self.reppts_out = nn.Conv2d(keypts_out_dim, reppts_out_dim, 1, 1, 0)

        bias_cls = bias_init_with_prob(0.01)
        if self.deform_conv:
            for k in _KERNELS:
                normal_init(getattr(self, 'cls_dfmconv_%d' % k), std=0.01)
                normal_init(getattr(self, 'keypts_dfmconv_%d' % k), std=0.01)
        else:
            normal_init(self.cls_conv, std=0.01)
            normal_init(self.keypts_conv, std=0.01)
        normal_init(self.cls_out, std=0.01, bias=bias_cls)
        normal_init(self.keypts_out, std=0.01)
        normal_init(self.reppts_out, std=0.01)

    def _dcn_offsets(self, reppts_offset, like):
        """per kernel size: gradient-scaled reppoints minus the regular grid (KP3:131-143)"""
        if (_FUSED_OFFSETS and reppts_offset.is_cuda and reppts_offset.dtype == torch.float32
                and not torch.is_autocast_enabled()):
            return list(_RepOffsets.apply(reppts_offset, float(self.gradient_mul)))     # one HIP pass each way (csrc/glue.hip)
        offsets, start = [], 0
        for k, n in zip(_KERNELS, _GROUP_POINTS):
            part = reppts_offset[:, 2 * start:2 * (start + n), :, :]
            start += n
            part = self.gradient_mul * part + (1 - self.gradient_mul) * part.detach()
            offsets.append(part - self._base_offset_on(k, like))
        return offsets

    def _base_offset_on(self, k, like):
        """the regular-grid tensor on ``like``'s device / dtype.  The reference keeps it as a plain CPU attribute
        (not a buffer, so not in the state_dict) and uploads it with type_as() on every call -- a blocking
        host->device copy, six per training step; here the upload happens once per device."""
        cache = self.__dict__.setdefault('_base_offset_cache', {})
        key = (k, like.device, like.dtype)
        if key not in cache:
            cache[key] = getattr(self, 'dcn_base_offset_%d' % k).to(device=like.device, dtype=like.dtype)
        return cache[key]

    def forward(self, cls_feat, pts_feat, reppts_offset=None):
        if self.deform_conv:
            offsets = self._dcn_offsets(reppts_offset, pts_feat)
            pads = [getattr(self, 'dcn_pad_%d' % k) for k in _KERNELS]
            cls_dfmconv_feat, keypts_dfmconv_feat = dcn.deform_conv_cat_multi(
                [cls_feat, pts_feat], offsets,
                [[getattr(self, 'cls_dfmconv_%d' % k).weight for k in _KERNELS],
                 [getattr(self, 'keypts_dfmconv_%d' % k).weight for k in _KERNELS]], pads)
            # (conv_bias_act: training takes the split MFMA kernels -- also for these 13 / 588 / 166-channel outputs, whose reductions
            # end inside a 16-channel chunk -- and falls back to conv_infer everywhere else)
            cls_out = conv1x1.conv_bias_act(self.cls_out, cls_dfmconv_feat)
            keypts_out = conv1x1.conv_bias_act(self.keypts_out, keypts_dfmconv_feat)
            reppts_out = conv1x1.conv_bias_act(self.reppts_out, keypts_out)
        else:
            cls_out = conv1x1.conv_bias_act(self.cls_out, conv1x1.conv_bias_act(self.cls_conv, cls_feat, relu=True))
            keypts_out = conv1x1.conv_bias_act(self.keypts_out, conv1x1.conv_bias_act(self.keypts_conv, pts_feat, relu=True))
            reppts_out = conv1x1.conv_bias_act(self.reppts_out, keypts_out)
        return cls_out, keypts_out, reppts_out


def _normalised_loss(loss_module, pred, target, weight, normalize_term, avg_factor):
    """``loss(pred / normalize_term, target / normalize_term, weight, avg_factor)`` (KP3:621-665); a SmoothL1Loss takes the
    divisor itself, so that its fused HIP op works on the raw tensors"""
    from .losses import SmoothL1Loss
    if type(loss_module) is SmoothL1Loss:
        return loss_module(pred, target, weight, avg_factor=avg_factor, divisor=normalize_term)
    return loss_module(pred / normalize_term, target / normalize_term, weight, avg_factor=avg_factor)


class PointHeadMixin(object):
    """point-set helpers shared by the KGDet head and the serial / parallel two-stage heads
    (identical code in KP3:342-410, 497-579 and reppoints_head_kp_serial.py:187-252, 341-423)"""

    def points2bbox(self, pts, y_first=True):
        """point set [B, 2n, H, W] -> box [B, 4, H, W] (x1, y1, x2, y2), KP3:342-391"""
        pts_reshape = pts.reshape(pts.shape[0], -1, 2, *pts.shape[2:])
        pts_y = pts_reshape[:, :, 0, ...] if y_first else pts_reshape[:, :, 1, ...]
        pts_x = pts_reshape[:, :, 1, ...] if y_first else pts_reshape[:, :, 0, ...]
        if self.transform_method == 'minmax':
            pass
        elif self.transform_method == 'partial_minmax':
            pts_y = pts_y[:, :4, ...]
            pts_x = pts_x[:, :4, ...]
        elif self.transform_method == 'moment':
            moment_transfer = (self.moment_transfer * self.moment_mul) + (
                self.moment_transfer.detach() * (1 - self.moment_mul))
            if pts.dim() == 2:  # [N, 2n] point sets (serial head loss): one "location" per row
                return moment.moment_bbox(pts.reshape(pts.shape[0], -1, 1, 1), moment_transfer, y_first).reshape(-1, 4)
            return moment.moment_bbox(pts, moment_transfer, y_first)
        else:
            raise NotImplementedError
        return torch.cat([pts_x.min(dim=1, keepdim=True)[0], pts_y.min(dim=1, keepdim=True)[0],
                          pts_x.max(dim=1, keepdim=True)[0], pts_y.max(dim=1, keepdim=True)[0]], dim=1)

    def points2kpt(self, pts, y_first=True):
        """(y, x) interleaved -> (x, y) interleaved channel order, KP3:393-410"""
        pts_reshape = pts.reshape(pts.shape[0], -1, 2, *pts.shape[2:])
        pts_y = pts_reshape[:, :, 0, ...] if y_first else pts_reshape[:, :, 1, ...]
        pts_x = pts_reshape[:, :, 1, ...] if y_first else pts_reshape[:, :, 0, ...]
        return torch.stack([pts_x, pts_y], dim=2).reshape(*pts.shape)

    def get_points(self, featmap_sizes, img_metas, device='cuda'):
        """grid centres and valid flags of every image and level (KP3:497-535)"""
        num_imgs = len(img_metas)
        num_levels = len(featmap_sizes)
        multi_level_points = [
            self.point_generators[i].grid_points(featmap_sizes[i], self.point_strides[i], device=device)
            for i in range(num_levels)
        ]
        points_list = [[point.clone() for point in multi_level_points] for _ in range(num_imgs)]
        valid_flag_list = []
        for img_meta in img_metas:
            multi_level_flags = []
            for i in range(num_levels):
                point_stride = self.point_strides[i]
                feat_h, feat_w = featmap_sizes[i]
                h, w, _ = img_meta['pad_shape']
                valid_feat_h = min(int(np.ceil(h / point_stride)), feat_h)
                valid_feat_w = min(int(np.ceil(w / point_stride)), feat_w)
                multi_level_flags.append(self.point_generators[i].valid_flags(
                    (feat_h, feat_w), (valid_feat_h, valid_feat_w), device=device))
            valid_flag_list.append(multi_level_flags)
        return points_list, valid_flag_list

    def centers_to_bboxes(self, point_list):
        """pseudo boxes of side point_base_scale*stride around the centres (MaxIoUAssigner only)"""
        bbox_list = []
        for point in point_list:
            bbox = []
            for i_lvl in range(len(self.point_strides)):
                scale = self.point_base_scale * self.point_strides[i_lvl] * 0.5
                bbox_shift = torch.Tensor([-scale, -scale, scale, scale]).view(1, 4).type_as(point[0])
                bbox_center = torch.cat([point[i_lvl][:, :2], point[i_lvl][:, :2]], dim=1)
                bbox.append(bbox_center + bbox_shift)
            bbox_list.append(bbox)
        return bbox_list

    def offset_to_pts(self, center_list, pred_list, y_first=True):
        """per level [B, H*W, 2n] image coordinates (x, y interleaved) = offset * stride + centre"""
        num_points = pred_list[0].size(1) // 2
        pts_list = []
        for i_lvl in range(len(self.point_strides)):
            pred = pred_list[i_lvl]                                       # [B, 2n, H, W]
            B = pred.shape[0]
            if (_FUSED_OFFSETS and pred.is_cuda and pred.dtype == torch.float32 and not torch.is_autocast_enabled()
                    and pred.numel() > 0):
                centers = torch.stack([center_list[i_img][i_lvl][:, :2] for i_img in range(B)], 0)
                pts_list.append(_PtsFromOffsets.apply(pred, centers, float(self.point_strides[i_lvl]), bool(y_first)))
                continue
            shift = pred.permute(0, 2, 3, 1).reshape(B, -1, num_points, 2)
            if y_first:
                shift = shift.flip(-1)                                    # (y, x) -> (x, y)
            centers = torch.stack([center_list[i_img][i_lvl][:, :2] for i_img in range(B)], 0)
            pts = shift * self.point_strides[i_lvl] + centers.unsqueeze(2)
            pts_list.append(pts.reshape(B, -1, 2 * num_points))
        return pts_list


@HEADS.register_module
class RepPointsHeadKp3RepCas1AssignOnce(PointHeadMixin, nn.Module):
    """Three-stage keypoint-guided RepPoints head (stage 1 plain, stages 2-3 deformable), one target
    assignment shared by all stages."""

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
                 flip_forward=False,
                 conv_cfg=None,
                 norm_cfg=None,
                 loss_cls_1=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=0.5),
                 loss_cls_2=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=0.5),
                 loss_cls_3=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                 loss_bbox_1=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=0.5),
                 loss_bbox_2=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=0.5),
                 loss_bbox_3=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0),
                 loss_kpt_1=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=0.5),
                 loss_kpt_2=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=0.5),
                 loss_kpt_3=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0),
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
        self.num_reppts = sum(_GROUP_POINTS)  # the config's num_reppts is ignored, as in KP3:258
        self.gradient_mul = gradient_mul
        self.point_base_scale = point_base_scale
        self.point_strides = point_strides
        self.flip_forward = flip_forward
        self.conv_cfg = conv_cfg
        self.norm_cfg = norm_cfg
        self.use_sigmoid_cls = loss_cls_3.get('use_sigmoid', False)
        self.sampling = loss_cls_3['type'] not in ['FocalLoss']
        for stage, cfgs in enumerate(((loss_cls_1, loss_bbox_1, loss_kpt_1), (loss_cls_2, loss_bbox_2, loss_kpt_2),
                                      (loss_cls_3, loss_bbox_3, loss_kpt_3)), 1):
            setattr(self, 'loss_cls_%d' % stage, build_loss(cfgs[0]))
            setattr(self, 'loss_bbox_%d' % stage, build_loss(cfgs[1]))
            setattr(self, 'loss_kpt_%d' % stage, build_loss(cfgs[2]))
        self.use_grid_points = use_grid_points
        self.center_init = center_init
        self.transform_method = transform_method
        if self.transform_method == 'moment':
            self.moment_transfer = nn.Parameter(data=torch.zeros(2), requires_grad=True)
            self.moment_mul = moment_mul
        self.cls_out_channels = self.num_classes - 1 if self.use_sigmoid_cls else self.num_classes
        self.point_generators = [PointGenerator() for _ in self.point_strides]
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
        for stage, deform in ((1, False), (2, True), (3, True)):
            setattr(self, 'kp_rep_block_%d' % stage,
                    Kp3RepBlock(deform, self.cls_out_channels, self.feat_channels, self.point_feat_channels,
                                self.num_reppts, self.num_keypts, self.gradient_mul))

    def init_weights(self):
        for m in self.cls_convs:
            normal_init(m.conv, std=0.01)
        for m in self.reg_convs:
            normal_init(m.conv, std=0.01)

    def forward_single(self, x):
        cls_feat = x
        pts_feat = x
        for cls_conv in self.cls_convs:
            cls_feat = cls_conv(cls_feat)
        for reg_conv in self.reg_convs:
            pts_feat = reg_conv(pts_feat)

        cls_out_1, keypts_out_1, reppts_out_1 = self.kp_rep_block_1(cls_feat, pts_feat)
        bbox_out_1 = self.points2bbox(reppts_out_1)

        cls_out_2, keypts_out_2, reppts_out_2 = self.kp_rep_block_2(cls_feat, pts_feat, reppts_out_1)
        keypts_out_2 = keypts_out_2 + keypts_out_1.detach()
        reppts_out_2 = reppts_out_2 + reppts_out_1.detach()
        bbox_out_2 = self.points2bbox(reppts_out_2)

        cls_out_3, keypts_out_3, reppts_out_3 = self.kp_rep_block_3(cls_feat, pts_feat, reppts_out_2)
        keypts_out_3 = keypts_out_3 + keypts_out_2.detach()
        reppts_out_3 = reppts_out_3 + reppts_out_2.detach()
        bbox_out_3 = self.points2bbox(reppts_out_3)
        return (cls_out_1, cls_out_2, cls_out_3, keypts_out_1, keypts_out_2, keypts_out_3, bbox_out_1,
                bbox_out_2, bbox_out_3)

    def forward_single_flip(self, feat, img_metas):
        """test-time horizontal-flip fusion of all nine maps (KP3:448-488)"""
        output = self.forward_single(feat)
        output_flip = self.forward_single(torch.flip(feat, [3]))
        num_stage = len(output) // 3
        flip_indices = img_metas[0]['flip_indices']
        fused = []
        for i in range(len(output)):
            back = torch.flip(output_flip[i], [3])
            kind = i // num_stage
            if kind == 1:      # keypoint offsets: (y, x) pairs -> negate x, swap left/right keypoints
                back[:, 1::2, :, :] = -back[:, 1::2, :, :]
                back = back[:, flip_indices, :, :]
            elif kind == 2:    # boxes (x1, y1, x2, y2): negate x and swap x1/x2
                back[:, 0::2, :, :] = -back[:, 0::2, :, :]
                back = back[:, [2, 1, 0, 3], :, :]
            fused.append((output[i] + back) / 2)
        return tuple(fused)

    def forward(self, feats, img_metas):
        if self.flip_forward:
            return multi_apply(self.forward_single_flip, feats, img_metas=img_metas)
        return multi_apply(self.forward_single, feats)

    def loss_single(self, cls_score_1, cls_score_2, cls_score_3, kpt_pred_1, kpt_pred_2, kpt_pred_3, bbox_pred_1,
                    bbox_pred_2, bbox_pred_3, labels, label_weights, bbox_gt, bbox_weights, kpt_gt, kpt_weights,
                    stride, num_total_samples):
        labels = labels.reshape(-1)
        label_weights = label_weights.reshape(-1)
        normalize_term = self.point_base_scale * stride
        bbox_gt = bbox_gt.reshape(-1, 4)
        bbox_weights = bbox_weights.reshape(-1, 4)
        # keypoint weights: visible keypoints of a positive share a total weight of 4 (KP3:641-644;
        # the reference normalises the target tensor in place)
        kpt_gt = kpt_gt.reshape(-1, self.num_keypts * 2)
        kpt_weights = kpt_weights.reshape(-1, self.num_keypts * 2)
        kpt_pos_num = kpt_weights.sum(1)
        # (rows without a visible keypoint are all zero, so dividing them by 1 instead of skipping them with a
        #  boolean mask gives the same tensor without a device->host round trip)
        kpt_weights = kpt_weights / kpt_pos_num.clamp(min=1).unsqueeze(1) * 4

        losses = []
        for stage, cls_score in enumerate((cls_score_1, cls_score_2, cls_score_3), 1):
            cls_score = cls_score.permute(0, 2, 3, 1).reshape(-1, self.cls_out_channels)
            losses.append(getattr(self, 'loss_cls_%d' % stage)(cls_score, labels, label_weights,
                                                               avg_factor=num_total_samples))
        for stage, bbox_pred in enumerate((bbox_pred_1, bbox_pred_2, bbox_pred_3), 1):
            losses.append(_normalised_loss(getattr(self, 'loss_bbox_%d' % stage), bbox_pred.reshape(-1, 4), bbox_gt,
                                           bbox_weights, normalize_term, num_total_samples))
        for stage, kpt_pred in enumerate((kpt_pred_1, kpt_pred_2, kpt_pred_3), 1):
            losses.append(_normalised_loss(getattr(self, 'loss_kpt_%d' % stage), kpt_pred.reshape(-1, self.num_keypts * 2),
                                           kpt_gt, kpt_weights, normalize_term, num_total_samples))
        return tuple(losses)

    def loss(self, cls_scores_1, cls_scores_2, cls_scores_3, keypts_preds_1, keypts_preds_2, keypts_preds_3,
             bbox_preds_1, bbox_preds_2, bbox_preds_3, gt_bboxes, gt_labels, gt_keypoints, img_metas, cfg,
             gt_bboxes_ignore=None):
        featmap_sizes = [featmap.size()[-2:] for featmap in cls_scores_3]
        assert len(featmap_sizes) == len(self.point_generators)
        label_channels = self.cls_out_channels if self.use_sigmoid_cls else 1
        device = cls_scores_3[0].device

        if cfg.uniform.assigner['type'] != 'PointAssigner':
            raise NotImplementedError
        # the part of every level's grid inside each image's own pad_shape (KP3:524-535; host arithmetic on the image
        # metas): a batch of mixed shapes leaves some grid points invalid in almost every step, and they take the SAME
        # sync-free path (round 6: valid extents in the dense targets and in the fused loss)
        valid_sizes = [[(min(int(np.ceil(meta['pad_shape'][0] / s)), fs[0]), min(int(np.ceil(meta['pad_shape'][1] / s)), fs[1]))
                        for s, fs in zip(self.point_strides, featmap_sizes)] for meta in img_metas]
        all_valid = all(v == tuple(fs) for per_img in valid_sizes for v, fs in zip(per_img, featmap_sizes))
        from . import head_loss
        stages = ((cls_scores_1, cls_scores_2, cls_scores_3), (keypts_preds_1, keypts_preds_2, keypts_preds_3),
                  (bbox_preds_1, bbox_preds_2, bbox_preds_3))
        level0 = None if all_valid else [per_img[0] for per_img in valid_sizes]
        if head_loss.applicable(self, cfg.uniform, stages[0], stages[1], stages[2], gt_bboxes, gt_labels, gt_keypoints,
                                gt_bboxes_ignore, valid_sizes=level0):
            # assignment + targets + the nine losses from the raw maps: four HIP launches (csrc/head_loss.hip)
            return head_loss.head_loss(self, cfg.uniform, stages[0], stages[1], stages[2], gt_bboxes, gt_labels,
                                       gt_keypoints, valid_sizes=level0)

        center_list, valid_flag_list = self.get_points(featmap_sizes, img_metas, device=device)
        kpt_coords = [self.offset_to_pts(center_list, p) for p in (keypts_preds_1, keypts_preds_2, keypts_preds_3)]
        bbox_coords = [self.offset_to_pts(center_list, p, y_first=False)
                       for p in (bbox_preds_1, bbox_preds_2, bbox_preds_3)]
        candidate_list = center_list
        if not self.sampling and dense_targets_applicable(cfg.uniform, len(self.point_strides), all_valid,
                                                          gt_bboxes_ignore):
            cls_reg_targets = point_target_kp_dense(candidate_list, gt_bboxes, gt_keypoints, cfg.uniform,
                                                    gt_labels_list=gt_labels,
                                                    valid_flag_list=None if all_valid else valid_flag_list)   # no host syncs
        else:
            cls_reg_targets = point_target_kp(candidate_list, valid_flag_list, gt_bboxes, gt_keypoints, img_metas,
                                              cfg.uniform, gt_bboxes_ignore_list=gt_bboxes_ignore,
                                              gt_labels_list=gt_labels, label_channels=label_channels,
                                              sampling=self.sampling)
        (labels_list, label_weights_list, bbox_gt_list, candidate_list, bbox_weights_list, keypoint_gt_list,
         keypoint_weights_list, num_total_pos, num_total_neg) = cls_reg_targets
        num_total_samples = (num_total_pos + num_total_neg if self.sampling else num_total_pos)

        per_level = multi_apply(self.loss_single, cls_scores_1, cls_scores_2, cls_scores_3, kpt_coords[0],
                                kpt_coords[1], kpt_coords[2], bbox_coords[0], bbox_coords[1], bbox_coords[2],
                                labels_list, label_weights_list, bbox_gt_list, bbox_weights_list, keypoint_gt_list,
                                keypoint_weights_list, self.point_strides, num_total_samples=num_total_samples)
        names = ['loss_cls_1', 'loss_cls_2', 'loss_cls_3', 'loss_bbox_1', 'loss_bbox_2', 'loss_bbox_3',
                 'loss_kpt_1', 'loss_kpt_2', 'loss_kpt_3']
        return dict(zip(names, per_level))

    # ------------------------------------------------------------------------------------------
    def _decode_level(self, cls_score, bbox_pred, kpt_pred, points, stride, img_shape, cfg):
        """one image, one level: scores [n, C], boxes [n, 4], keypoints [n, K, 3] in image coordinates"""
        num_kpt = self.num_keypts
        num_kp_channel = kpt_pred.size(0) // num_kpt
        assert num_kp_channel == 2 or num_kp_channel == 3
        assert cls_score.size()[-2:] == bbox_pred.size()[-2:] == kpt_pred.size()[-2:]
        cls_score = cls_score.permute(1, 2, 0).reshape(-1, self.cls_out_channels)
        scores = cls_score.sigmoid() if self.use_sigmoid_cls else cls_score.softmax(-1)
        bbox_pred = bbox_pred.permute(1, 2, 0).reshape(-1, 4)
        if num_kp_channel == 3:
            kpt_pred = kpt_pred.permute(1, 2, 0).reshape(-1, num_kpt * num_kp_channel)
        else:  # visibility not predicted: pad with 1
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
        bboxes = bbox_pred * stride + bbox_pos_center
        kpts = kpt_pred.view(-1, num_kpt, 3).clone()
        kpts[:, :, :2] = kpts[:, :, :2] * stride + points[:, :2].unsqueeze(dim=1)
        # clamp to img_shape itself, not img_shape - 1 (KP3:882-888)
        bboxes = torch.stack([bboxes[:, 0].clamp(min=0, max=img_shape[1]), bboxes[:, 1].clamp(min=0, max=img_shape[0]),
                              bboxes[:, 2].clamp(min=0, max=img_shape[1]), bboxes[:, 3].clamp(min=0, max=img_shape[0])],
                             dim=-1)
        kpts[:, :, 0] = kpts[:, :, 0].clamp(min=0, max=img_shape[1])
        kpts[:, :, 1] = kpts[:, :, 1].clamp(min=0, max=img_shape[0])
        return bboxes, scores, kpts

    def get_bboxes_single(self, cls_scores, bbox_preds, kpt_preds, mlvl_points, img_shape, scale_factor, cfg,
                          rescale=False, nms=True):
        assert len(cls_scores) == len(bbox_preds) == len(mlvl_points) == len(kpt_preds)
        decoded = [self._decode_level(c, b, k, p, self.point_strides[i], img_shape, cfg)
                   for i, (c, b, k, p) in enumerate(zip(cls_scores, bbox_preds, kpt_preds, mlvl_points))]
        mlvl_bboxes = torch.cat([d[0] for d in decoded])
        mlvl_scores = torch.cat([d[1] for d in decoded])
        mlvl_kpts = torch.cat([d[2] for d in decoded])
        if rescale:
            # (a scalar factor divides directly; new_tensor() is a blocking host->device upload per image)
            sf = scale_factor if isinstance(scale_factor, (int, float)) else mlvl_bboxes.new_tensor(scale_factor)
            mlvl_bboxes /= sf
            mlvl_kpts[:, :, 0:2] = mlvl_kpts[:, :, 0:2] / sf
            mlvl_kpts = mlvl_kpts.reshape(-1, self.num_keypts * 3)
        if self.use_sigmoid_cls:
            padding = mlvl_scores.new_zeros(mlvl_scores.shape[0], 1)
            mlvl_scores = torch.cat([padding, mlvl_scores], dim=1)
        if nms:
            return multiclass_nms_kp(mlvl_bboxes, mlvl_scores, mlvl_kpts, cfg.score_thr, cfg.nms, cfg.max_per_img)
        return mlvl_bboxes, mlvl_scores, mlvl_kpts

    def get_bboxes(self, cls_scores_1, cls_scores_2, cls_scores_3, keypts_preds_1, keypts_preds_2, keypts_preds_3,
                   bbox_preds_1, bbox_preds_2, bbox_preds_3, img_metas, cfg, rescale=False, nms=True):
        """detections from the final stage.  With hard NMS the whole batch is suppressed in one launch."""
        # decode in fp32 whatever the convolutions ran in: bf16 cannot hold a pixel coordinate (8-bit mantissa)
        cls_score_final = [t.float() for t in cls_scores_3]
        bbox_preds = [t.float() for t in bbox_preds_3]
        keypts_preds_final = [t.float() for t in keypts_preds_3]
        assert len(cls_score_final) == len(keypts_preds_final) == len(bbox_preds)
        kpt_preds = [self.points2kpt(keypts_pred) for keypts_pred in keypts_preds_final]
        num_levels = len(cls_score_final)
        device = cls_score_final[0].device
        mlvl_points = [
            self.point_generators[i].grid_points(cls_score_final[i].size()[-2:], self.point_strides[i], device=device)
            for i in range(num_levels)
        ]
        if nms and self._packed_ok(cls_score_final, img_metas, cfg):
            det, label, kp, count = self.get_bboxes_packed(cls_score_final, bbox_preds, kpt_preds, mlvl_points,
                                                           img_metas, cfg, rescale)
            counts = count.tolist()                      # the only device->host read of the post-processing
            return [(det[b, :n], label[b, :n], kp[b, :n]) for b, n in enumerate(counts)]
        per_image = []
        for img_id in range(len(img_metas)):
            cls_score_list = [cls_score_final[i][img_id].detach() for i in range(num_levels)]
            bbox_pred_list = [bbox_preds[i][img_id].detach() for i in range(num_levels)]
            kpt_pred_list = [kpt_preds[i][img_id].detach() for i in range(num_levels)]
            img_shape = img_metas[img_id]['img_shape']
            scale_factor = img_metas[img_id]['scale_factor']
            batched = nms and cfg.nms.get('type', 'nms') == 'nms' and cls_score_final[0].is_cuda
            per_image.append(self.get_bboxes_single(cls_score_list, bbox_pred_list, kpt_pred_list, mlvl_points,
                                                    img_shape, scale_factor, cfg, rescale, nms and not batched))
        if not (nms and cfg.nms.get('type', 'nms') == 'nms' and cls_score_final[0].is_cuda):
            return per_image
        n_max = max(p[0].shape[0] for p in per_image)

        def pad(t):
            return t if t.shape[0] == n_max else torch.cat([t, t.new_zeros((n_max - t.shape[0], ) + t.shape[1:])])

        return multiclass_nms_kp_batched(torch.stack([pad(p[0]) for p in per_image]),
                                         torch.stack([pad(p[1]) for p in per_image]),
                                         torch.stack([pad(p[2]) for p in per_image]), cfg.score_thr, cfg.nms,
                                         cfg.max_per_img)

    def get_bboxes_packed_tensor(self, cls_scores_1, cls_scores_2, cls_scores_3, keypts_preds_1, keypts_preds_2,
                                 keypts_preds_3, bbox_preds_1, bbox_preds_2, bbox_preds_3, img_metas, cfg,
                                 rescale=False):
        """The batch's detections as ONE device tensor [B, max_per_img, 7 + 3K] -- columns: box (4), score, label,
        count (repeated), landmarks -- produced without any host read (capturable in a HIP graph); None when the
        packed path does not apply.  ``unpack_results`` splits its host copy per image."""
        cls3, box3 = [t.float() for t in cls_scores_3], [t.float() for t in bbox_preds_3]
        if not self._packed_ok(cls3, img_metas, cfg):
            return None
        kpt3 = [self.points2kpt(t.float()) for t in keypts_preds_3]
        points = [self.point_generators[i].grid_points(cls3[i].size()[-2:], self.point_strides[i], device=cls3[i].device)
                  for i in range(len(cls3))]
        det, label, kp, count = self.get_bboxes_packed(cls3, box3, kpt3, points, img_metas, cfg, rescale)
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
        """``get_bboxes`` with the results on the host as numpy arrays; on the packed path the whole batch comes
        back in ONE device->host copy."""
        packed = self.get_bboxes_packed_tensor(*args, **kwargs)
        if packed is None:
            return [(d.float().cpu().numpy(), lab.cpu().numpy(), k.float().cpu().numpy())
                    for d, lab, k in self.get_bboxes(*args, **kwargs)]
        return self.unpack_results(packed.cpu().numpy())

    # ------------------------------------------------------------------------------------------
    # whole-batch decode + fused NMS: no per-image Python loop, no host read before the results
    # ------------------------------------------------------------------------------------------
    def _packed_ok(self, cls_scores, img_metas, cfg):
        if not (cls_scores[0].is_cuda and cfg.nms.get('type', 'nms') == 'nms' and self.use_sigmoid_cls
                and cfg.max_per_img > 0):
            return False
        if not all(isinstance(m['scale_factor'], (int, float)) for m in img_metas):
            return False
        n = sum(min(c.shape[-2] * c.shape[-1], cfg.get('nms_pre', -1)) if cfg.get('nms_pre', -1) > 0
                else c.shape[-2] * c.shape[-1] for c in cls_scores)
        return n <= 4096 and n * self.cls_out_channels <= 16384 and self.cls_out_channels <= 64

    @staticmethod
    def _per_image(values, device):
        """python numbers, one per image -> a scalar when they agree, else a [B, 1] tensor uploaded without blocking"""
        if all(v == values[0] for v in values):
            return values[0]
        return torch.tensor(values, dtype=torch.float32).pin_memory().to(device, non_blocking=True).unsqueeze(1)

    def _decode_level_batch(self, cls_score, bbox_pred, kpt_pred, points, stride, lim_w, lim_h, cfg):
        """``_decode_level`` for all images at once: scores [B,n,C], boxes [B,n,4], landmarks [B,n,K,3]"""
        B, num_kpt = cls_score.shape[0], self.num_keypts
        ch = kpt_pred.size(1) // num_kpt
        assert ch == 2 or ch == 3
        scores = cls_score.permute(0, 2, 3, 1).reshape(B, -1, self.cls_out_channels).sigmoid()
        bbox_pred = bbox_pred.permute(0, 2, 3, 1).reshape(B, -1, 4)
        kpt_pred = kpt_pred.permute(0, 2, 3, 1).reshape(B, -1, num_kpt, ch)
        if ch == 2:
            kpt_pred = torch.cat([kpt_pred, kpt_pred.new_ones(kpt_pred[..., :1].shape)], dim=-1)
        ctr = points[:, :2].unsqueeze(0).expand(B, -1, -1)
        nms_pre = cfg.get('nms_pre', -1)
        if nms_pre > 0 and scores.shape[1] > nms_pre:
            _, top = scores.max(dim=2)[0].topk(nms_pre, dim=1)
            ctr = torch.gather(ctr, 1, top.unsqueeze(-1).expand(B, nms_pre, 2))
            bbox_pred = torch.gather(bbox_pred, 1, top.unsqueeze(-1).expand(B, nms_pre, 4))
            kpt_pred = torch.gather(kpt_pred, 1, top.view(B, nms_pre, 1, 1).expand(B, nms_pre, num_kpt, 3))
            scores = torch.gather(scores, 1, top.unsqueeze(-1).expand(B, nms_pre, scores.shape[2]))
        bboxes = bbox_pred * stride + torch.cat([ctr, ctr], dim=2)
        kpts = kpt_pred.clone()
        kpts[..., :2] = kpts[..., :2] * stride + ctr.unsqueeze(2)

        def clamp(t, lim):      # to [0, img_shape] (not img_shape - 1, KP3:882-888)
            if isinstance(lim, (int, float)):
                return t.clamp(min=0, max=lim)
            return torch.minimum(t.clamp(min=0), lim.view((B, ) + (1, ) * (t.dim() - 1)))

        bboxes = torch.stack([clamp(bboxes[..., 0], lim_w), clamp(bboxes[..., 1], lim_h), clamp(bboxes[..., 2], lim_w),
                              clamp(bboxes[..., 3], lim_h)], dim=-1)
        kpts[..., 0] = clamp(kpts[..., 0], lim_w)
        kpts[..., 1] = clamp(kpts[..., 1], lim_h)
        return bboxes, scores, kpts

    def get_bboxes_packed(self, cls_scores, bbox_preds, kpt_preds, mlvl_points, img_metas, cfg, rescale=False):
        """final-stage maps -> fixed-size device tensors (det [B,M,5], labels [B,M], landmarks [B,M,3K], count [B])"""
        device = cls_scores[0].device
        lim_w = self._per_image([float(m['img_shape'][1]) for m in img_metas], device)
        lim_h = self._per_image([float(m['img_shape'][0]) for m in img_metas], device)
        decoded = [self._decode_level_batch(cls_scores[i].detach().float(), bbox_preds[i].detach().float(),
                                            kpt_preds[i].detach().float(), mlvl_points[i], self.point_strides[i],
                                            lim_w, lim_h, cfg) for i in range(len(cls_scores))]
        bboxes = torch.cat([d[0] for d in decoded], dim=1)
        scores = torch.cat([d[1] for d in decoded], dim=1)
        kpts = torch.cat([d[2] for d in decoded], dim=1)
        if rescale:
            sf = self._per_image([float(m['scale_factor']) for m in img_metas], device)
            if isinstance(sf, float):
                bboxes = bboxes / sf
                kpts[..., 0:2] = kpts[..., 0:2] / sf
            else:   # torch divides by a python scalar as x * (1 / s) in fp32: do the same per image
                inv = torch.reciprocal(sf)
                bboxes = bboxes * inv.view(-1, 1, 1)
                kpts[..., 0:2] = kpts[..., 0:2] * inv.view(-1, 1, 1, 1)
        from .postprocess import multiclass_nms_kp_fused
        return multiclass_nms_kp_fused(bboxes, scores, kpts.reshape(kpts.shape[0], kpts.shape[1], -1), cfg.score_thr,
                                       float(cfg.nms['iou_thr']), cfg.max_per_img)


HEADS.register_alias('KGDetHead', RepPointsHeadKp3RepCas1AssignOnce)
