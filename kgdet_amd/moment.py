"""Fused moment bounding box (``points2bbox`` with ``transform_method='moment'``).

Reference: mmdet/models/anchor_heads/reppoints_head_kp3rep_cas_1_assign_once.py:342-391 --
about ten element-wise / reduction ops per call there, one HIP kernel each way here
(csrc/moment.hip).  ``moment_transfer`` is passed already blended with its detached copy
(``t * mul + t.detach() * (1 - mul)``, :378-379) so the 0.01 gradient scale stays in autograd.
"""
import ctypes

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib


class MomentBBoxFunction(Function):

    @staticmethod
    def forward(ctx, pts, moment_transfer, y_first=True):
        if not pts.is_cuda:
            raise NotImplementedError('moment_bbox has no CPU implementation')
        B, C2, H, W = pts.shape
        assert C2 % 2 == 0
        pts = pts.contiguous().float()
        mt = moment_transfer.contiguous().float()
        bbox = pts.new_empty(B, 4, H, W)
        _lib.check(_lib.lib().kgdet_moment_bbox_forward(
            _lib.ptr(pts), _lib.ptr(mt), ctypes.c_int32(B), ctypes.c_int32(C2 // 2), ctypes.c_int32(H * W),
            ctypes.c_int32(1 if y_first else 0), _lib.ptr(bbox), _lib.current_stream()),
            'kgdet_moment_bbox_forward')
        ctx.save_for_backward(pts, mt)
        ctx.y_first = y_first
        return bbox

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_bbox):
        pts, mt = ctx.saved_tensors
        B, C2, H, W = pts.shape
        grad_bbox = grad_bbox.contiguous().float()
        grad_pts = torch.empty_like(pts)
        grad_mt = torch.empty_like(mt)        # (written by the kernel: per-block partials added in block order, no atomics)
        L = _lib.lib()
        ws_bytes = L.kgdet_moment_bbox_backward_workspace_bytes(ctypes.c_int32(B), ctypes.c_int32(H * W))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=pts.device)
        _lib.check(L.kgdet_moment_bbox_backward(
            _lib.ptr(pts), _lib.ptr(mt), _lib.ptr(grad_bbox), ctypes.c_int32(B), ctypes.c_int32(C2 // 2),
            ctypes.c_int32(H * W), ctypes.c_int32(1 if ctx.y_first else 0), _lib.ptr(grad_pts),
            _lib.ptr(grad_mt), _lib.ptr(ws), ctypes.c_size_t(ws_bytes), _lib.current_stream()), 'kgdet_moment_bbox_backward')
        return grad_pts, grad_mt, None


moment_bbox = MomentBBoxFunction.apply
