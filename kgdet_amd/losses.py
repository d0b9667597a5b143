"""Loss modules of the KGDet head: ``FocalLoss`` (on the HIP sigmoid-focal op) and
``SmoothL1Loss``, with the reference's weighting / avg_factor reduction.

Mirrors mmdet/models/losses/focal_loss.py:28-82, smooth_l1_loss.py:8-45, utils.py:7-52.
"""
import ctypes
import functools
import os as _os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import focal_loss as _focal_op
from .registry import LOSSES


def reduce_loss(loss, reduction):
    reduction_enum = F._Reduction.get_enum(reduction)
    if reduction_enum == 0:
        return loss
    elif reduction_enum == 1:
        return loss.mean()
    elif reduction_enum == 2:
        return loss.sum()


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        loss = reduce_loss(loss, reduction)
    else:
        if reduction == 'mean':
            loss = loss.sum() / avg_factor
        elif reduction != 'none':
            raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


def weighted_loss(loss_func):

    @functools.wraps(loss_func)
    def wrapper(pred, target, weight=None, reduction='mean', avg_factor=None, **kwargs):
        loss = loss_func(pred, target, **kwargs)
        return weight_reduce_loss(loss, weight, reduction, avg_factor)

    return wrapper


def sigmoid_focal_loss(pred, target, weight=None, gamma=2.0, alpha=0.25, reduction='mean', avg_factor=None):
    loss = _focal_op.sigmoid_focal_loss(pred, target, gamma, alpha)
    if weight is not None:
        weight = weight.view(-1, 1)
    return weight_reduce_loss(loss, weight, reduction, avg_factor)


@LOSSES.register_module
class FocalLoss(nn.Module):

    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super(FocalLoss, self).__init__()
        assert use_sigmoid is True, 'Only sigmoid focal loss supported now.'
        self.use_sigmoid = use_sigmoid
        self.gamma = gamma
        self.alpha = alpha
        self.reduction = reduction
        self.loss_weight = loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        if not self.use_sigmoid:
            raise NotImplementedError
        return self.loss_weight * sigmoid_focal_loss(pred, target, weight, gamma=self.gamma, alpha=self.alpha,
                                                     reduction=reduction, avg_factor=avg_factor)


@weighted_loss
def smooth_l1_loss(pred, target, beta=1.0):
    assert beta > 0
    assert pred.size() == target.size() and target.numel() > 0
    diff = torch.abs(pred - target)
    return torch.where(diff < beta, 0.5 * diff * diff / beta, diff - 0.5 * beta)


FUSED_SMOOTH_L1 = _os.environ.get('KGDET_FUSED_SMOOTH_L1', '1') == '1'    # csrc/smooth_l1.hip (0: the reference's torch chain; A/B)


def fused_smooth_l1_applicable(pred, target, weight, reduction, avg_factor):
    return (FUSED_SMOOTH_L1 and pred.is_cuda and pred.dtype == torch.float32 and target.dtype == torch.float32
            and pred.shape == target.shape and pred.numel() > 0 and not target.requires_grad
            and (weight is None or (weight.dtype == torch.float32 and weight.shape == pred.shape and not weight.requires_grad))
            and reduction == 'mean' and avg_factor is not None and not torch.is_autocast_enabled())


class _SmoothL1Sum(torch.autograd.Function):
    """sum(weight * smooth_l1(pred / d - target / d)) as one HIP pass each way (csrc/smooth_l1.hip); the reference runs it as
    ~14 element-wise kernels forward and as many backward (smooth_l1_loss.py:8-19, utils.py:7-52)"""

    @staticmethod
    def forward(ctx, pred, target, weight, beta, divisor):
        from . import _lib
        L = _lib.lib()
        partial = torch.empty(L.kgdet_smooth_l1_partials(), dtype=torch.float32, device=pred.device)
        out = torch.empty((), dtype=torch.float32, device=pred.device)
        _lib.check(L.kgdet_smooth_l1_sum_forward(
            _lib.ptr(pred), _lib.ptr(target), _lib.ptr(weight) if weight is not None else None, ctypes.c_int64(pred.numel()),
            ctypes.c_float(beta), ctypes.c_float(divisor), _lib.ptr(partial), _lib.ptr(out), _lib.current_stream()),
            'smooth_l1_sum_forward')
        ctx.save_for_backward(pred, target, weight)
        ctx.beta, ctx.divisor = beta, divisor
        return out

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        pred, target, weight = ctx.saved_tensors
        grad = torch.empty_like(pred)
        g = g.contiguous().float()
        _lib.check(_lib.lib().kgdet_smooth_l1_sum_backward(
            _lib.ptr(pred), _lib.ptr(target), _lib.ptr(weight) if weight is not None else None, _lib.ptr(g),
            ctypes.c_int64(pred.numel()), ctypes.c_float(ctx.beta), ctypes.c_float(ctx.divisor), _lib.ptr(grad),
            _lib.current_stream()), 'smooth_l1_sum_backward')
        return grad, None, None, None, None


@LOSSES.register_module
class SmoothL1Loss(nn.Module):

    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0):
        super(SmoothL1Loss, self).__init__()
        self.beta = beta
        self.reduction = reduction
        self.loss_weight = loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, divisor=None, **kwargs):
        """``divisor`` (not in the reference signature): the loss of ``pred / divisor`` and ``target / divisor`` -- the head
        normalises both by ``point_base_scale * stride`` (KP3:621-665); passing the divisor instead lets the fused HIP op
        take the raw tensors.  Same value either way."""
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        if fused_smooth_l1_applicable(pred, target, weight, reduction, avg_factor):
            s = _SmoothL1Sum.apply(pred.contiguous(), target.contiguous(), None if weight is None else weight.contiguous(),
                                   float(self.beta), 1.0 if divisor is None else float(divisor))
            return self.loss_weight * (s / avg_factor)
        if divisor is not None:
            pred, target = pred / divisor, target / divisor
        return self.loss_weight * smooth_l1_loss(pred, target, weight, beta=self.beta, reduction=reduction,
                                                 avg_factor=avg_factor, **kwargs)
