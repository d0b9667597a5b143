"""Plain-PyTorch building blocks around the hot path: conv + norm + activation module, norm
factory and the weight initialisers the reference takes from mmcv.

Mirrors mmdet/models/utils/conv_module.py:44-164 (``ConvModule``; child names ``conv`` / ``gn`` /
``bn`` are part of the checkpoint-key contract), mmdet/models/utils/norm.py:3-55 and
mmdet/models/utils/weight_init.py / mmcv.cnn init helpers.  Dense convolutions stay on
PyTorch-ROCm (MIOpen); nothing here is hand-written HIP.
"""
import ctypes
import os as _os
import warnings

import numpy as np
import torch
import torch.nn as nn

from . import conv1x1

_NORMS = {'BN': ('bn', nn.BatchNorm2d), 'SyncBN': ('bn', nn.SyncBatchNorm), 'GN': ('gn', nn.GroupNorm)}
_CONVS = {'Conv': nn.Conv2d}


def build_norm_layer(cfg, num_features, postfix=''):
    assert isinstance(cfg, dict) and 'type' in cfg
    cfg_ = dict(cfg)
    layer_type = cfg_.pop('type')
    if layer_type not in _NORMS:
        raise KeyError('Unrecognized norm type {}'.format(layer_type))
    abbr, norm_layer = _NORMS[layer_type]
    assert isinstance(postfix, (int, str))
    name = abbr + str(postfix)
    requires_grad = cfg_.pop('requires_grad', True)
    cfg_.setdefault('eps', 1e-5)
    if layer_type != 'GN':
        layer = norm_layer(num_features, **cfg_)
    else:
        assert 'num_groups' in cfg_
        layer = norm_layer(num_channels=num_features, **cfg_)
    for param in layer.parameters():
        param.requires_grad = requires_grad
    return name, layer


def build_conv_layer(cfg, *args, **kwargs):
    if cfg is None:
        cfg_ = dict(type='Conv')
    else:
        assert isinstance(cfg, dict) and 'type' in cfg
        cfg_ = dict(cfg)
    layer_type = cfg_.pop('type')
    if layer_type not in _CONVS:
        raise KeyError('Unrecognized norm type {}'.format(layer_type))
    return _CONVS[layer_type](*args, **kwargs, **cfg_)


def constant_init(module, val, bias=0):
    nn.init.constant_(module.weight, val)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def xavier_init(module, gain=1, bias=0, distribution='normal'):
    assert distribution in ['uniform', 'normal']
    if distribution == 'uniform':
        nn.init.xavier_uniform_(module.weight, gain=gain)
    else:
        nn.init.xavier_normal_(module.weight, gain=gain)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def normal_init(module, mean=0, std=1, bias=0):
    nn.init.normal_(module.weight, mean, std)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def kaiming_init(module, mode='fan_out', nonlinearity='relu', bias=0, distribution='normal'):
    assert distribution in ['uniform', 'normal']
    if distribution == 'uniform':
        nn.init.kaiming_uniform_(module.weight, mode=mode, nonlinearity=nonlinearity)
    else:
        nn.init.kaiming_normal_(module.weight, mode=mode, nonlinearity=nonlinearity)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def bias_init_with_prob(prior_prob):
    """initialise a conv/fc bias so that sigmoid(bias) == prior_prob"""
    return float(-np.log((1 - prior_prob) / prior_prob))


FUSED_GN = _os.environ.get('KGDET_FUSED_GN', '1') == '1'     # csrc/group_norm.hip (0: nn.GroupNorm + nn.ReLU; A/B)
GN_SPLIT = _os.environ.get('KGDET_GN_SPLIT', '1') == '1'     # groups beyond 65536 elements on the sliced kernels (0: ATen; A/B)


def gn_act_applicable(x, norm):
    return (FUSED_GN and type(norm) is nn.GroupNorm and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and x.is_contiguous() and x.numel() > 0 and not torch.is_autocast_enabled()
            and norm.num_channels // norm.num_groups <= 64
            and (norm.num_channels // norm.num_groups) * x.shape[2] * x.shape[3] < (2 ** 31 if GN_SPLIT else 65537))


def _gn_scratch(x, N, C, groups, HW):
    from . import _lib
    L = _lib.lib()
    L.kgdet_gn_act_scratch_floats.restype = ctypes.c_size_t
    n = L.kgdet_gn_act_scratch_floats(ctypes.c_int64(N), ctypes.c_int32(C), ctypes.c_int32(groups), ctypes.c_int64(HW))
    return torch.empty(n, dtype=torch.float32, device=x.device) if n else None


def gn_act_bf16_applicable(x, norm):
    """inference under autocast: the convolution in front hands over bf16; torch would cast to fp32, normalise in three
    launches, apply the ReLU and cast back in front of the next convolution"""
    return (FUSED_GN and type(norm) is nn.GroupNorm and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4
            and (x.is_contiguous() or x.is_contiguous(memory_format=torch.channels_last))
            and x.numel() > 0 and not torch.is_grad_enabled()
            and norm.num_channels // norm.num_groups <= 64
            and (norm.num_channels // norm.num_groups) * x.shape[2] * x.shape[3] < (2 ** 31 if GN_SPLIT else 65537))


def gn_act_bf16(x, norm, relu):
    from . import _lib
    N, C = x.shape[0], x.shape[1]
    y = torch.empty_like(x)      # (x's layout: a channels-last tower stays channels-last)
    if (C // norm.num_groups) * x.shape[2] * x.shape[3] > 65536:
        # the largest level of a five-level head (8 channels x 16800 pixels per group): the split kernels (round 4; torch's
        # path here was RowwiseMoments + two element-wise passes + casts per layer: ~0.25 ms of a config-5 batch each)
        scratch = _gn_scratch(x, N, C, norm.num_groups, x.shape[2] * x.shape[3])
        _lib.check(_lib.lib().kgdet_gn_act_forward_bf16_split(
            _lib.ptr(x), ctypes.c_int32(0 if x.is_contiguous() else 1), _lib.ptr(norm.weight), _lib.ptr(norm.bias),
            ctypes.c_int32(norm.num_groups), ctypes.c_float(norm.eps), ctypes.c_int32(1 if relu else 0), _lib.ptr(y),
            _lib.ptr(scratch), ctypes.c_int64(N), ctypes.c_int32(C), ctypes.c_int64(x.shape[2] * x.shape[3]),
            _lib.current_stream()), 'gn_act_forward_bf16_split')
        return y
    _lib.check(_lib.lib().kgdet_gn_act_forward_bf16(
        _lib.ptr(x), ctypes.c_int32(0 if x.is_contiguous() else 1), _lib.ptr(norm.weight), _lib.ptr(norm.bias), ctypes.c_int32(norm.num_groups), ctypes.c_float(norm.eps),
        ctypes.c_int32(1 if relu else 0), _lib.ptr(y), ctypes.c_int64(N), ctypes.c_int32(C),
        ctypes.c_int64(x.shape[2] * x.shape[3]), _lib.current_stream()), 'gn_act_forward_bf16')
    return y


class _GNAct(torch.autograd.Function):
    """``[relu](group_norm(x))`` (conv_module.py:142-165: norm, then activate) on csrc/group_norm.hip"""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, relu):
        from . import _lib
        N, C = x.shape[0], x.shape[1]
        HW = x.shape[2] * x.shape[3]
        y = torch.empty_like(x)
        stats = torch.empty((2, N * groups), dtype=torch.float32, device=x.device)
        scratch = _gn_scratch(x, N, C, groups, HW)     # (large groups: slice moments, csrc/group_norm.hip)
        _lib.check(_lib.lib().kgdet_gn_act_forward_split(
            _lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), ctypes.c_int32(groups), ctypes.c_float(eps),
            ctypes.c_int32(1 if relu else 0), _lib.ptr(y), ctypes.c_void_p(stats[0].data_ptr()),
            ctypes.c_void_p(stats[1].data_ptr()), _lib.ptr(scratch), ctypes.c_int64(N), ctypes.c_int32(C), ctypes.c_int64(HW),
            _lib.current_stream()), 'gn_act_forward')
        ctx.save_for_backward(x, y if relu else None, gamma, stats)
        ctx.groups, ctx.relu, ctx.has_beta = groups, relu, beta is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        from . import _lib
        x, y, gamma, stats = ctx.saved_tensors
        N, C = x.shape[0], x.shape[1]
        HW = x.shape[2] * x.shape[3]
        gy = gy.contiguous()
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dgb = torch.empty((2, N, C), dtype=torch.float32, device=x.device)
        scratch = _gn_scratch(x, N, C, ctx.groups, HW)
        _lib.check(_lib.lib().kgdet_gn_act_backward_split(
            _lib.ptr(gy), _lib.ptr(x), _lib.ptr(y), _lib.ptr(gamma), ctypes.c_void_p(stats[0].data_ptr()),
            ctypes.c_void_p(stats[1].data_ptr()), ctypes.c_int32(ctx.groups), ctypes.c_int32(1 if ctx.relu else 0),
            _lib.ptr(gx), _lib.ptr(dgb), _lib.ptr(scratch), ctypes.c_int64(N), ctypes.c_int32(C), ctypes.c_int64(HW),
            _lib.current_stream()), 'gn_act_backward')
        sums = dgb.sum(1) if N > 1 else dgb[:, 0]
        ggamma = sums[0] if (gamma is not None and ctx.needs_input_grad[1]) else None
        gbeta = sums[1] if (ctx.has_beta and ctx.needs_input_grad[2]) else None
        return gx, ggamma, gbeta, None, None, None


def gn_act(x, norm, relu):
    return _GNAct.apply(x, norm.weight, norm.bias, norm.num_groups, norm.eps, relu)


class ConvModule(nn.Module):
    """conv -> norm -> activation, in configurable order."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias='auto', conv_cfg=None, norm_cfg=None, activation='relu', inplace=True,
                 order=('conv', 'norm', 'act')):
        super(ConvModule, self).__init__()
        assert conv_cfg is None or isinstance(conv_cfg, dict)
        assert norm_cfg is None or isinstance(norm_cfg, dict)
        self.conv_cfg = conv_cfg
        self.norm_cfg = norm_cfg
        self.activation = activation
        self.inplace = inplace
        self.order = order
        assert isinstance(self.order, tuple) and len(self.order) == 3
        assert set(order) == set(['conv', 'norm', 'act'])

        self.with_norm = norm_cfg is not None
        self.with_activatation = activation is not None
        if bias == 'auto':
            bias = False if self.with_norm else True
        self.with_bias = bias
        if self.with_norm and self.with_bias:
            warnings.warn('ConvModule has norm and bias at the same time')

        self.conv = build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size, stride=stride,
                                     padding=padding, dilation=dilation, groups=groups, bias=bias)
        for attr in ('in_channels', 'out_channels', 'kernel_size', 'stride', 'padding', 'dilation',
                     'transposed', 'output_padding', 'groups'):
            setattr(self, attr, getattr(self.conv, attr))

        if self.with_norm:
            norm_channels = out_channels if order.index('norm') > order.index('conv') else in_channels
            self.norm_name, norm = build_norm_layer(norm_cfg, norm_channels)
            self.add_module(self.norm_name, norm)

        if self.with_activatation:
            if self.activation not in ['relu']:
                raise ValueError('{} is currently not supported.'.format(self.activation))
            self.activate = nn.ReLU(inplace=inplace)

        self.init_weights()

    @property
    def norm(self):
        return getattr(self, self.norm_name)

    def init_weights(self):
        nonlinearity = 'relu' if self.activation is None else self.activation
        kaiming_init(self.conv, nonlinearity=nonlinearity)
        if self.with_norm:
            constant_init(self.norm, 1, bias=0)

    def forward(self, x, activate=True, norm=True):
        fused_act = False
        for li, layer in enumerate(self.order):
            if layer == 'conv':
                conv = self.conv
                if (type(conv) is nn.Conv2d and conv.bias is None and torch.is_grad_enabled()
                        and conv1x1.applicable(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups)):
                    x = conv1x1.conv_split(x, conv.weight)     # fp32 training: split-operand MFMA kernels (conv1x1.hip)
                elif (type(conv) is nn.Conv2d and conv.bias is None and torch.is_grad_enabled()
                        and conv1x1.odd_map_applicable(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups)):
                    x = conv1x1.unpad_odd(conv1x1.conv_split(conv1x1.pad_odd(x), conv.weight), x.shape[3])
                elif (type(conv) is nn.Conv2d and torch.is_grad_enabled()
                        and conv1x1.applicable_stride2(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups)):
                    # (the FPN's extra levels: 3x3 stride 2.  MIOpen's fp32 forward for the 13 x 21 and 7 x 11 outputs splits K with
                    #  float atomics -- every tensor computed on those two levels differed in its last bits from run to run)
                    x = conv1x1.conv3x3_stride2(x, conv.weight)
                    if conv.bias is not None:
                        x = x + conv.bias.view(1, -1, 1, 1)
                elif type(conv) is nn.Conv2d and conv.bias is not None and torch.is_grad_enabled():
                    x = conv1x1.conv_bias_act(conv, x, relu=False)   # (config 5's FPN: biased convolutions without a norm)
                else:
                    x = conv1x1.conv_infer(conv, x)
            elif layer == 'norm' and norm and self.with_norm:
                relu_next = (activate and self.with_activatation and li + 1 < len(self.order)
                             and self.order[li + 1] == 'act')
                if gn_act_applicable(x, self.norm):
                    x = gn_act(x, self.norm, relu_next)     # GroupNorm (+ the ReLU that follows) as one HIP pass each way
                    fused_act = relu_next
                elif gn_act_bf16_applicable(x, self.norm):
                    x = gn_act_bf16(x, self.norm, relu_next)
                    fused_act = relu_next
                else:
                    x = self.norm(x)
            elif layer == 'act' and activate and self.with_activatation:
                if not fused_act:
                    x = self.activate(x)
        return x



# ------------------------------------------------------------------------------------------------
# A module shared by several pyramid levels as ONE autograd node
# ------------------------------------------------------------------------------------------------
class _SharedLevels(torch.autograd.Function):
    """``run(l, x_l, params)`` for every level l, recorded as ONE node of the outer graph.  The reference applies its head to the
    five FPN levels one after the other (reppoints_head_kp_serial.py:495-497 multi_apply): every shared parameter then receives
    five gradients, which the autograd engine adds pairwise as they arrive -- 235 `add` launches (1.1 ms) of a config-5 step.  Here
    each level runs on leaf ALIASES of the parameters (same storage) under its own recorded sub-graph; backward differentiates the
    five sub-graphs and sums the parameter gradients with multi-tensor adds: four launch groups for all parameters together.  The sum
    runs level 0 + level 1 + ... in that order for every parameter: deterministic."""

    @staticmethod
    def forward(ctx, run, n_levels, *flat):
        xs, params = flat[:n_levels], flat[n_levels:]
        ctx.levels, ctx.n_params = [], len(params)
        outs_flat = []
        for l, x in enumerate(xs):
            with torch.enable_grad():
                xin = x.detach().requires_grad_(x.requires_grad)
                proxies = [p.detach().requires_grad_(p.requires_grad) for p in params]
                outs = tuple(run(l, xin, proxies))
            ctx.levels.append((xin, proxies, outs))
            outs_flat.extend(o.detach() for o in outs)
        ctx.n_out = len(outs_flat) // max(n_levels, 1)
        return tuple(outs_flat)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gouts):
        gxs, acc = [], [None] * ctx.n_params
        for l, (xin, proxies, outs) in enumerate(ctx.levels):
            gl = gouts[l * ctx.n_out:(l + 1) * ctx.n_out]
            sel = [(o, g) for o, g in zip(outs, gl) if g is not None and o.requires_grad]
            inputs = ([xin] if xin.requires_grad else []) + [p for p in proxies if p.requires_grad]
            if not sel or not inputs:
                gxs.append(None)
                continue
            grads = torch.autograd.grad([o for o, _ in sel], inputs, [g for _, g in sel], allow_unused=True)
            grads = list(grads)
            gxs.append(grads.pop(0) if xin.requires_grad else None)
            it = iter(grads)
            mine, theirs = [], []
            for i, p in enumerate(proxies):
                if not p.requires_grad:
                    continue
                g = next(it)
                if g is None:
                    continue
                if acc[i] is None:
                    acc[i] = g if g.is_contiguous() else g.contiguous()
                else:
                    mine.append(acc[i])
                    theirs.append(g)
            if mine:
                torch._foreach_add_(mine, theirs)      # (multi-tensor: a few launches for all parameters of the level)
        ctx.levels = None
        return (None, None) + tuple(gxs) + tuple(acc)


def shared_levels(module, single, xs):
    """``[single(x) for x in xs]`` transposed (as multi_apply) with ``module``'s parameters shared through ``_SharedLevels``"""
    from torch.nn.utils import stateless
    names, params = zip(*[(n, p) for n, p in module.named_parameters()])

    def run(l, x, proxies):
        for q, p in zip(proxies, params):
            conv1x1.alias(q, p)
        with stateless._reparametrize_module(module, dict(zip(names, proxies))):
            return single(x)
    flat = _SharedLevels.apply(run, len(xs), *xs, *params)
    n_out = len(flat) // len(xs)
    return tuple([flat[l * n_out + j] for l in range(len(xs))] for j in range(n_out))
