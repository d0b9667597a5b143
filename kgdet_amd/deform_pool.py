"""Deformable PS-RoI pooling -- mirror of ``mmdet/ops/dcn/deform_pool.py``
(DeformRoIPoolingFunction :9-69, DeformRoIPooling :75-103, DeformRoIPoolingPack :106-164,
ModulatedDeformRoIPoolingPack :167-245) on the HIP kernels in csrc/psroi.hip (LDS-staged RoI windows, deterministic tile-gather backward).
"""
import ctypes

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib


def _shape(data, rois, offset, spatial_scale, out_size, out_channels, no_trans, group_size, part_size,
           sample_per_part, trans_std):
    s = _lib.PsroiShape()
    s.B, s.C, s.H, s.W = data.shape
    s.R = rois.shape[0]
    s.out_dim = out_channels
    s.group_size = group_size
    s.pooled_size = out_size
    s.part_size = part_size
    s.sample_per_part = sample_per_part
    s.no_trans = 1 if no_trans else 0
    s.num_classes = 1 if no_trans else offset.shape[1] // 2
    s.spatial_scale = spatial_scale
    s.trans_std = trans_std
    return s


class DeformRoIPoolingFunction(Function):

    @staticmethod
    def forward(ctx, data, rois, offset, spatial_scale, out_size, out_channels, no_trans, group_size=1,
                part_size=None, sample_per_part=4, trans_std=.0):
        ctx.spatial_scale = spatial_scale
        ctx.out_size = out_size
        ctx.out_channels = out_channels
        ctx.no_trans = no_trans
        ctx.group_size = group_size
        ctx.part_size = out_size if part_size is None else part_size
        ctx.sample_per_part = sample_per_part
        ctx.trans_std = trans_std

        assert 0.0 <= ctx.trans_std <= 1.0
        if not data.is_cuda:
            raise NotImplementedError

        data = data.contiguous().float()       # the C ABI is float32 only (autocast activations are cast up)
        rois = rois.contiguous().float()
        offset = offset.contiguous().float()
        n = rois.shape[0]
        output = data.new_empty(n, out_channels, out_size, out_size)
        output_count = data.new_empty(n, out_channels, out_size, out_size)
        shape = _shape(data, rois, offset, spatial_scale, out_size, out_channels, no_trans, group_size,
                       ctx.part_size, sample_per_part, trans_std)
        L = _lib.lib()
        ws_bytes = L.kgdet_deform_psroi_forward_workspace_bytes(ctypes.byref(shape))
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=data.device)     # the cell-major copy of the map
        _lib.check(L.kgdet_deform_psroi_forward(
            ctypes.byref(shape), _lib.ptr(data), _lib.ptr(rois), None if no_trans else _lib.ptr(offset),
            _lib.ptr(output), _lib.ptr(output_count), _lib.ptr(ws), ctypes.c_size_t(ws_bytes), _lib.current_stream()),
            'kgdet_deform_psroi_forward')
        ctx.shape = shape
        ctx.save_for_backward(data, rois, offset)
        ctx.output_count = output_count
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        if not grad_output.is_cuda:
            raise NotImplementedError
        data, rois, offset = ctx.saved_tensors
        grad_output = grad_output.contiguous().float()
        # both gradients are written in full by the kernels (a gather per output tile / one workgroup per
        # (RoI, class): no atomics, so no zero-filled accumulators -- deform_pool.py:57-59 zero-fills for atomicAdd)
        grad_input = torch.empty_like(data)
        grad_rois = None
        grad_offset = torch.empty_like(offset)
        L = _lib.lib()
        ws_bytes = L.kgdet_deform_psroi_backward_workspace_bytes(ctypes.byref(ctx.shape))
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=data.device)
        _lib.check(L.kgdet_deform_psroi_backward(
            ctypes.byref(ctx.shape), _lib.ptr(grad_output), _lib.ptr(ctx.output_count), _lib.ptr(data),
            _lib.ptr(rois), None if ctx.no_trans else _lib.ptr(offset), _lib.ptr(grad_input),
            None if ctx.no_trans else _lib.ptr(grad_offset), _lib.ptr(ws), ctypes.c_size_t(ws_bytes),
            _lib.current_stream()), 'kgdet_deform_psroi_backward')
        return (grad_input, grad_rois, grad_offset, None, None, None, None, None, None, None, None)


deform_roi_pooling = DeformRoIPoolingFunction.apply


class DeformRoIPooling(nn.Module):

    def __init__(self, spatial_scale, out_size, out_channels, no_trans, group_size=1, part_size=None,
                 sample_per_part=4, trans_std=.0):
        super(DeformRoIPooling, self).__init__()
        self.spatial_scale = spatial_scale
        self.out_size = out_size
        self.out_channels = out_channels
        self.no_trans = no_trans
        self.group_size = group_size
        self.part_size = out_size if part_size is None else part_size
        self.sample_per_part = sample_per_part
        self.trans_std = trans_std

    def _pool(self, data, rois, offset, no_trans):
        return deform_roi_pooling(data, rois, offset, self.spatial_scale, self.out_size, self.out_channels,
                                  no_trans, self.group_size, self.part_size, self.sample_per_part,
                                  self.trans_std)

    def forward(self, data, rois, offset):
        if self.no_trans:
            offset = data.new_empty(0)
        return self._pool(data, rois, offset, self.no_trans)


def _fc_stack(in_features, hidden, n_layers, out_features):
    """n_layers Linear layers with ReLU between them; the last one zero-initialised."""
    seq = []
    ic = in_features
    for i in range(n_layers):
        oc = hidden if i < n_layers - 1 else out_features
        seq.append(nn.Linear(ic, oc))
        ic = oc
        if i < n_layers - 1:
            seq.append(nn.ReLU(inplace=True))
    return seq


class DeformRoIPoolingPack(DeformRoIPooling):

    def __init__(self, spatial_scale, out_size, out_channels, no_trans, group_size=1, part_size=None,
                 sample_per_part=4, trans_std=.0, num_offset_fcs=3, deform_fc_channels=1024):
        super(DeformRoIPoolingPack, self).__init__(spatial_scale, out_size, out_channels, no_trans,
                                                   group_size, part_size, sample_per_part, trans_std)
        self.num_offset_fcs = num_offset_fcs
        self.deform_fc_channels = deform_fc_channels

        if not no_trans:
            seq = _fc_stack(self.out_size * self.out_size * self.out_channels, self.deform_fc_channels,
                            self.num_offset_fcs, self.out_size * self.out_size * 2)
            self.offset_fc = nn.Sequential(*seq)
            self.offset_fc[-1].weight.data.zero_()
            self.offset_fc[-1].bias.data.zero_()

    def forward(self, data, rois):
        assert data.size(1) == self.out_channels
        if self.no_trans:
            return self._pool(data, rois, data.new_empty(0), True)
        n = rois.shape[0]
        x = self._pool(data, rois, data.new_empty(0), True)
        offset = self.offset_fc(x.view(n, -1))
        offset = offset.view(n, 2, self.out_size, self.out_size)
        return self._pool(data, rois, offset, False)


class ModulatedDeformRoIPoolingPack(DeformRoIPooling):

    def __init__(self, spatial_scale, out_size, out_channels, no_trans, group_size=1, part_size=None,
                 sample_per_part=4, trans_std=.0, num_offset_fcs=3, num_mask_fcs=2, deform_fc_channels=1024):
        super(ModulatedDeformRoIPoolingPack, self).__init__(spatial_scale, out_size, out_channels, no_trans,
                                                            group_size, part_size, sample_per_part, trans_std)
        self.num_offset_fcs = num_offset_fcs
        self.num_mask_fcs = num_mask_fcs
        self.deform_fc_channels = deform_fc_channels

        if not no_trans:
            in_f = self.out_size * self.out_size * self.out_channels
            self.offset_fc = nn.Sequential(*_fc_stack(in_f, self.deform_fc_channels, self.num_offset_fcs,
                                                      self.out_size * self.out_size * 2))
            self.offset_fc[-1].weight.data.zero_()
            self.offset_fc[-1].bias.data.zero_()
            seq = _fc_stack(in_f, self.deform_fc_channels, self.num_mask_fcs, self.out_size * self.out_size)
            seq.append(nn.Sigmoid())
            self.mask_fc = nn.Sequential(*seq)
            self.mask_fc[-2].weight.data.zero_()
            self.mask_fc[-2].bias.data.zero_()

    def forward(self, data, rois):
        assert data.size(1) == self.out_channels
        if self.no_trans:
            return self._pool(data, rois, data.new_empty(0), True)
        n = rois.shape[0]
        x = self._pool(data, rois, data.new_empty(0), True)
        offset = self.offset_fc(x.view(n, -1))
        offset = offset.view(n, 2, self.out_size, self.out_size)
        mask = self.mask_fc(x.view(n, -1))
        mask = mask.view(n, 1, self.out_size, self.out_size)
        return self._pool(data, rois, offset, False) * mask
