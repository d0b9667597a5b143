"""Sigmoid focal loss op -- mirror of ``mmdet/ops/sigmoid_focal_loss/sigmoid_focal_loss.py``
(SigmoidFocalLossFunction :8-32, sigmoid_focal_loss :35, SigmoidFocalLoss :39-54) on the HIP
kernels in csrc/focal.hip.  As in the reference there is no CPU path.
"""
import ctypes

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib


class SigmoidFocalLossFunction(Function):

    @staticmethod
    def forward(ctx, input, target, gamma=2.0, alpha=0.25):
        if not input.is_cuda:
            raise NotImplementedError('SigmoidFocalLoss is not implemented on the CPU')
        ctx.save_for_backward(input, target)
        num_classes = input.shape[1]
        ctx.num_classes = num_classes
        ctx.gamma = gamma
        ctx.alpha = alpha
        logits = input.contiguous().float()
        target = target.contiguous().long()
        loss = torch.empty_like(logits)
        _lib.check(_lib.lib().kgdet_sigmoid_focal_loss_forward(
            _lib.ptr(logits), _lib.ptr(target), ctypes.c_int64(logits.shape[0]), ctypes.c_int32(num_classes),
            ctypes.c_float(gamma), ctypes.c_float(alpha), _lib.ptr(loss), _lib.current_stream()),
            'kgdet_sigmoid_focal_loss_forward')
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, d_loss):
        input, target = ctx.saved_tensors
        logits = input.contiguous().float()
        target = target.contiguous().long()
        d_loss = d_loss.contiguous().float()
        d_input = torch.empty_like(logits)
        _lib.check(_lib.lib().kgdet_sigmoid_focal_loss_backward(
            _lib.ptr(logits), _lib.ptr(target), _lib.ptr(d_loss), ctypes.c_int64(logits.shape[0]),
            ctypes.c_int32(ctx.num_classes), ctypes.c_float(ctx.gamma), ctypes.c_float(ctx.alpha),
            _lib.ptr(d_input), _lib.current_stream()), 'kgdet_sigmoid_focal_loss_backward')
        return d_input, None, None, None


sigmoid_focal_loss = SigmoidFocalLossFunction.apply


class SigmoidFocalLoss(nn.Module):

    def __init__(self, gamma, alpha):
        super(SigmoidFocalLoss, self).__init__()
        self.gamma = gamma
        self.alpha = alpha

    def forward(self, logits, targets):
        assert logits.is_cuda
        loss = sigmoid_focal_loss(logits, targets, self.gamma, self.alpha)
        return loss.sum()

    def __repr__(self):
        return self.__class__.__name__ + '(gamma={}, alpha={})'.format(self.gamma, self.alpha)
