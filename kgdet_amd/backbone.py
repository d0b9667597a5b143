"""ResNet backbone (plain PyTorch-ROCm convolutions; MIOpen supplies the MFMA kernels).

Mirrors mmdet/models/backbones/resnet.py: ``BasicBlock`` :13-83, ``Bottleneck`` :86-265 (including
the optional deformable ``conv2`` + ``conv2_offset`` of the dcn configs, :162-186, 231-238),
``make_res_layer`` :268-328, ``ResNet`` :331-525.  Parameter names follow the checkpoint-key
contract (``conv1``, ``bn1``, ``layer{1..4}.{i}.conv{1,2,3}|bn{1,2,3}|downsample.{0,1}``).
BatchNorm stays in eval mode (``norm_eval``) and ``frozen_stages`` freezes the stem + first stages.
"""
import ctypes
import os as _os
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.modules.batchnorm import _BatchNorm

from . import conv1x1
from . import dcn as dcn_ops
from .layers import build_conv_layer, build_norm_layer, constant_init, kaiming_init
from .registry import BACKBONES


STEM_CONV = _os.environ.get('KGDET_STEM_CONV', '1') == '1'    # csrc/conv1x1.hip stem_conv7x7_s2 (0: MIOpen; A/B)
_stem_cache = {}


def _stem_conv(conv, x):
    """conv1 of the stem, no autograd (the caller checked that nothing here trains): the 7x7 / stride 2 / 3 -> 64 layer on the
    split-bf16 MFMA kernel (its packed weight cached per weight version), anything else through the module"""
    w = conv.weight
    if (STEM_CONV and x.is_contiguous() and tuple(w.shape) == (64, 3, 7, 7) and conv.stride == (2, 2) and conv.padding == (3, 3)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None and w.dtype == torch.float32):
        key = (w.data_ptr(), w._version)
        hit = _stem_cache.get(id(conv))
        # (id, address and version can all repeat once a module is freed and another built -- CPython reuses ids, the caching
        #  allocator addresses: the entry also holds a weak reference to the weight it was packed from)
        if hit is None or hit[0] != key or hit[2]() is not w:
            w160 = torch.zeros((64, 160), dtype=torch.float32, device=w.device)
            w160[:, :147] = w.detach().reshape(64, 147)
            hit = (key, conv1x1._pack(w160.view(64, 160, 1, 1), False), weakref.ref(w))
            _stem_cache[id(conv)] = hit
        from . import _lib
        B, _, H, W = x.shape
        y = torch.empty((B, 64, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().kgdet_stem_conv7x7_s2_fmt(
            _lib.ptr(hit[1]), _lib.ptr(x), _lib.ptr(y), ctypes.c_int64(B), ctypes.c_int32(H), ctypes.c_int32(W),
            ctypes.c_int32(1 if getattr(hit[1], 'kgdet_f16', False) else 0), _lib.current_stream()), 'stem_conv7x7_s2')
        return y
    return conv(x).contiguous()


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style='pytorch', with_cp=False,
                 conv_cfg=None, norm_cfg=dict(type='BN'), dcn=None, gcb=None, gen_attention=None):
        super(BasicBlock, self).__init__()
        assert dcn is None, 'Not implemented yet.'
        assert gen_attention is None, 'Not implemented yet.'
        assert gcb is None, 'Not implemented yet.'
        self.norm1_name, norm1 = build_norm_layer(norm_cfg, planes, postfix=1)
        self.norm2_name, norm2 = build_norm_layer(norm_cfg, planes, postfix=2)
        self.conv1 = build_conv_layer(conv_cfg, inplanes, planes, 3, stride=stride, padding=dilation,
                                      dilation=dilation, bias=False)
        self.add_module(self.norm1_name, norm1)
        self.conv2 = build_conv_layer(conv_cfg, planes, planes, 3, padding=1, bias=False)
        self.add_module(self.norm2_name, norm2)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride
        self.dilation = dilation
        assert not with_cp

    @property
    def norm1(self):
        return getattr(self, self.norm1_name)

    @property
    def norm2(self):
        return getattr(self, self.norm2_name)

    def forward(self, x):
        identity = x
        out = self.relu(self.norm1(self.conv1(x)))
        out = self.norm2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out += identity
        return self.relu(out)


_FUSE_EPI = _os.environ.get('KGDET_FUSE_EPI', '1') == '1'   # fp32 inference: epilogue inside conv_nn's store (0: separate pass, for A/B)
_fold_cache = {}   # id(conv) -> (weakref to conv, folded weight, folded bias)


def clear_fold_cache():
    _fold_cache.clear()
    _stem_cache.clear()


def _epilogue_(y, bias, residual, relu):
    """y = [relu](y + bias[c] [+ residual]) in place: one HIP pass (csrc/epilogue.hip) on the GPU, for NCHW and for
    channels-last storage."""
    if y.is_cuda and y.dtype in (torch.float32, torch.bfloat16) and y.dim() == 4:
        N, C = y.shape[0], y.shape[1]
        if y.is_contiguous():
            fmt = torch.contiguous_format
        elif y.is_contiguous(memory_format=torch.channels_last) and C % (4 if y.dtype == torch.float32 else 8) == 0:
            fmt = torch.channels_last
        else:
            fmt = None
        if fmt is not None and (residual is None or (residual.dtype == y.dtype and residual.shape == y.shape
                                                     and residual.is_contiguous(memory_format=fmt))):
            from . import _lib
            _lib.check(_lib.lib().kgdet_bias_act(
                _lib.ptr(y), _lib.ptr(bias), _lib.ptr(residual), ctypes.c_int64(N), ctypes.c_int32(C),
                ctypes.c_int64(y.numel() // max(N * C, 1)), ctypes.c_int32(0 if y.dtype == torch.float32 else 1),
                ctypes.c_int32(1 if relu else 0), ctypes.c_int32(1 if fmt is torch.channels_last else 0),
                _lib.current_stream()), 'bias_act')
            return y
    if bias is not None:
        y = y + bias.to(y.dtype).view(1, -1, 1, 1)
    if residual is not None:
        y = y + residual
    return F.relu(y, inplace=True) if relu else y


_BN = None
_bn_partials = {}


def _bn_lib():
    """bn_act entry points with argtypes declared once (plain ints / pointers marshal faster than c_* objects)"""
    global _BN
    if _BN is None:
        from . import _lib
        L = _lib.lib()
        vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
        L.kgdet_bn_act_partials.restype, L.kgdet_bn_act_partials.argtypes = ctypes.c_int32, [i64, i32, i64]
        L.kgdet_bn_act_forward.restype = ctypes.c_int
        L.kgdet_bn_act_forward.argtypes = [vp, vp, vp, vp, vp, f32, vp, vp, i64, i32, i64, i32, vp]
        L.kgdet_bn_act_backward.restype = ctypes.c_int
        L.kgdet_bn_act_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, f32, i32, i32, vp, vp, vp, vp, i64, i32, i64, vp]
        L.kgdet_bn_relu_maxpool.restype = ctypes.c_int
        L.kgdet_bn_relu_maxpool.argtypes = [vp, vp, vp, vp, vp, f32, vp, i64, i32, i32, i32, vp]
        _BN = L
    return _BN


def _p(t):
    return t.data_ptr() if t is not None else None


def _bn_act_forward(x, gamma, beta, mean, var, eps, residual, relu):
    from . import _lib
    N, C = x.shape[0], x.shape[1]
    HW = x.numel() // max(N * C, 1)
    y = torch.empty_like(x)
    _lib.check(_bn_lib().kgdet_bn_act_forward(
        _p(x), _p(gamma), _p(beta), _p(mean), _p(var), eps, _p(residual), _p(y), N, C, HW, 1 if relu else 0,
        _lib.raw_stream(x.device.index)), 'bn_act_forward')
    return y


def _bn_act_backward(gy, x, y, gamma, beta, mean, var, eps, has_res, relu, need_gx):
    """-> (grad_x or None, grad_residual-or-None (None: it IS gy), [grad_beta, grad_gamma] sums [2, C])"""
    from . import _lib
    N, C = x.shape[0], x.shape[1]
    HW = x.numel() // max(N * C, 1)
    L = _bn_lib()
    P = _bn_partials.get((N, C, HW))
    if P is None:
        P = _bn_partials[(N, C, HW)] = L.kgdet_bn_act_partials(N, C, HW)
    partial = torch.empty((2, C, max(P, 1)), dtype=torch.float32, device=x.device)
    gx = torch.empty_like(x) if need_gx else None
    masked = has_res and relu
    gres = torch.empty_like(x) if masked else None
    sums = torch.empty((2, C), dtype=torch.float32, device=x.device)
    _lib.check(L.kgdet_bn_act_backward(
        _p(gy), _p(x), _p(y), _p(gamma), _p(beta), _p(mean), _p(var), eps, 1 if has_res else 0, 1 if relu else 0,
        _p(gx), _p(gres), _p(partial), _p(sums), N, C, HW, _lib.raw_stream(x.device.index)), 'bn_act_backward')
    return gx, gres, sums


class _FrozenBNAct(torch.autograd.Function):
    """``[relu](batch_norm_eval(x) [+ residual])`` as one HIP pass each way (csrc/bn_act.hip)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mean, var, eps, residual, relu):
        y = _bn_act_forward(x, gamma, beta, mean, var, eps, residual, relu)
        ctx.eps, ctx.relu, ctx.has_res = eps, relu, residual is not None
        ctx.save_for_backward(x, y if (relu and residual is not None) else None, gamma, beta, mean, var)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, gamma, beta, mean, var = ctx.saved_tensors
        gy = gy.contiguous()
        gx, gres, sums = _bn_act_backward(gy, x, y, gamma, beta, mean, var, ctx.eps, ctx.has_res, ctx.relu,
                                          ctx.needs_input_grad[0])
        ggamma = sums[1] if (gamma is not None and ctx.needs_input_grad[1]) else None
        gbeta = sums[0] if (beta is not None and ctx.needs_input_grad[2]) else None
        if not ctx.has_res or not ctx.needs_input_grad[6]:
            gres = None
        elif gres is None:
            gres = gy
        return gx, ggamma, gbeta, None, None, None, gres, None


class _ConvBNAct(torch.autograd.Function):
    """``[relu](batch_norm_eval(conv(x, w)) [+ residual])`` as ONE autograd node: split-bf16 convolution kernels +
    the fused BatchNorm pass.  Same kernels as ``conv_split`` followed by ``frozen_bn_act``; merging the two nodes
    takes ~50 Python autograd-node round trips off the step, whose host side is as long as its GPU side.

    ``skip=True`` also returns x itself (an alias): a bottleneck takes its identity branch from there, so x has ONE
    consumer in the graph and the identity branch's gradient arrives HERE -- it is added in the store of the
    grad_input kernel instead of by an autograd accumulation pass over the whole activation (12 passes of up to 69 MB
    per step at 6.6 TB/s: 0.3 ms)."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, mean, var, eps, residual, relu, skip=False):
        weight = weight.contiguous()
        img, ctx.img_t = conv1x1.forward_images(x, weight)
        y = conv1x1._apply(img, x, weight.shape[0], weight.shape[2] * weight.shape[3])
        z = _bn_act_forward(y, gamma, beta, mean, var, eps, residual, relu)
        ctx.eps, ctx.relu, ctx.has_res = eps, relu, residual is not None
        ctx.save_for_backward(x, weight, y, z if (relu and residual is not None) else None, gamma, beta, mean, var)
        return (z, x) if skip else z

    @staticmethod
    def backward(ctx, gz, gskip=None):
        x, weight, y, z, gamma, beta, mean, var = ctx.saved_tensors
        gz = gz.contiguous()
        need_conv = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        gy, gres, sums = _bn_act_backward(gz, y, z, gamma, beta, mean, var, ctx.eps, ctx.has_res, ctx.relu, need_conv)
        if gskip is not None and (gskip.dtype != torch.float32 or not gskip.is_contiguous()):
            gskip = gskip.float().contiguous()
        gx = conv1x1.grad_input(weight, ctx.img_t, gy, residual=gskip) if ctx.needs_input_grad[0] else None
        gw = conv1x1.grad_weight(x, weight, gy) if ctx.needs_input_grad[1] else None
        ggamma = sums[1] if (gamma is not None and ctx.needs_input_grad[2]) else None
        gbeta = sums[0] if (beta is not None and ctx.needs_input_grad[3]) else None
        if not ctx.has_res or not ctx.needs_input_grad[7]:
            gres = None
        elif gres is None:
            gres = gz
        return gx, gw, ggamma, gbeta, None, None, None, gres, None, None


def _bn_fold_lib():
    L = _bn_lib()
    if not hasattr(L, '_kgdet_fold_ready'):
        vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
        L.kgdet_bn_fold_backward.restype = ctypes.c_int
        L.kgdet_bn_fold_backward.argtypes = [vp, vp, i32, vp, vp, i64, i32, i64, vp]
        L.kgdet_bn_fold_finish.restype = ctypes.c_int
        L.kgdet_bn_fold_finish.argtypes = [vp, i32, vp, vp, vp, vp, vp, f32, vp, vp, i32, i32, vp]
        L._kgdet_fold_ready = True
    return L


NHWC_EXTENDED = _os.environ.get('KGDET_NHWC_EXTENDED', '1') == '1'
RAW_BRANCHES = _os.environ.get('KGDET_RAW_BRANCHES', '1') == '1'   # 0: downsample / stride-2 conv2 take their own bias pass at bf16 inference (A/B)
GATE_FUSION = _os.environ.get('KGDET_GATE_FUSION', '1') == '1'    # 0: every ReLU node masks its own gradient (A/B)


class _GateLink(object):
    """Hand-over between two ``_ConvBNActFold`` nodes ``z = relu(...)`` -> ``u = conv(z)`` where u's node is the ONLY consumer of z
    (inside a bottleneck: conv1 -> conv2 -> conv3; between the bottlenecks of a layer: block output -> next block's conv1, whose
    ``skip`` alias also carries the identity branch).  The consumer's backward zeroes grad_z where z <= 0 in the store of its
    grad_input kernel (``conv1x1.grad_input(gate=z)``) and sets ``gated``; the producer's backward, which runs later, then skips
    its masking pass (read gradient, read z, write masked gradient: three passes over the activation, `relu_sum_bwd_kernel<true>`)
    and only sums the gradient per channel.  The mask is idempotent, so a producer whose link was never set (the consumer took
    another route) masks as before: nothing depends on the two nodes agreeing in advance."""
    __slots__ = ('gated',)

    def __init__(self):
        self.gated = False


class _ConvBNActFold(torch.autograd.Function):
    """``[relu](batch_norm_eval(conv(x, w)) [+ residual])`` with the BatchNorm FOLDED into the convolution: the step scope's
    pack launch wrote the operand images of ``w * s`` (s = gamma / sqrt(var + eps)), so the forward is ONE kernel --
    ``z = [relu](conv(x, w s) + t [+ r])`` in the convolution's store, t = beta - mean s -- and y = conv(x, w) is neither
    written nor kept for the backward.  Backward: ``g = gz [z > 0]`` (one pass, which also IS the residual branch's gradient),
    grad_x = conv_grad_input(g, w s), G = conv_grad_weight(x, g), grad_w = s G, grad_beta = sum g and
    grad_gamma = invstd (<w[o], G[o]> - mean sum g): sum_p g[o, p] y[o, p] = <w[o], G[o]> exactly, so neither y nor a
    non-zero gamma is needed (mmdet zero-initialises the last BatchNorm of every bottleneck).  Replaces _ConvBNAct from a
    pair's second step on (conv1x1.fold_images); same arguments and results."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, mean, var, eps, residual, relu, skip, fold, gate_in=None, gate_out=None):
        # gate_in: the _GateLink through which this node's output gradient may arrive already masked by [z > 0];
        # gate_out: the link of the node that produced x = relu(.) -- this node masks grad_x with [x > 0] and says so
        img, ctx.img_t, s, t = fold
        z = conv1x1._apply(img, x, weight.shape[0], weight.shape[2] * weight.shape[3], 1, t, residual, relu)
        ctx.eps, ctx.relu, ctx.has_res = eps, relu, residual is not None
        ctx.gate_in = gate_in if relu else None
        ctx.gate_out = gate_out if (gate_out is not None and conv1x1.gate_applicable(x.shape)) else None
        ctx.save_for_backward(x, weight, z if relu else None, s, mean, var)
        return (z, x) if skip else z

    @staticmethod
    def backward(ctx, gz, gskip=None):
        from . import _lib
        x, weight, z, s, mean, var = ctx.saved_tensors
        gz = gz.contiguous()
        L = _bn_fold_lib()
        N, O = gz.shape[0], gz.shape[1]
        HW = gz.numel() // max(N * O, 1)
        P = _bn_partials.get((N, O, HW))
        if P is None:
            P = _bn_partials[(N, O, HW)] = L.kgdet_bn_act_partials(N, O, HW)
        mask = ctx.relu and not (ctx.gate_in is not None and ctx.gate_in.gated)    # (gated: the consumer of z masked gz already)
        need_w, need_g, need_b = ctx.needs_input_grad[1], ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        # gz is final (no mask to apply here): no pass over it at all when the weight-gradient kernel can sum its rows
        sums_in_wgrad = (not mask and GATE_FUSION and (need_w or need_g) and conv1x1.grad_weight_fold_route(x, weight) != 0)
        g = torch.empty_like(gz) if mask else gz
        st = _lib.raw_stream(gz.device.index)
        partial = None
        if not sums_in_wgrad:
            partial = torch.empty((O, max(P, 1)), dtype=torch.float32, device=gz.device)
            _lib.check(L.kgdet_bn_fold_backward(_p(gz), _p(z) if mask else None, 1 if mask else 0, _p(g) if mask else None,
                                                _p(partial), N, O, HW, st), 'bn_fold_backward')
        if gskip is not None and (gskip.dtype != torch.float32 or not gskip.is_contiguous()):
            gskip = gskip.float().contiguous()
        gx = None
        if ctx.needs_input_grad[0]:
            gate = x if (ctx.gate_out is not None and GATE_FUSION) else None
            gx = conv1x1.grad_input(weight, ctx.img_t, g, residual=gskip, gate=gate)
            if gate is not None:
                ctx.gate_out.gated = True
        gw = sums = None
        if need_w or need_g:      # the split sum of the weight gradient, grad_w = s G and the BatchNorm sums in one launch
            both = conv1x1.grad_weight_fold(x, weight, g, s, mean, var, ctx.eps, partial, max(P, 1))
            assert both is not None or partial is not None
            if both is not None:
                gw, sums = both
            else:
                gw = conv1x1.grad_weight(x, weight, g)
        if sums is None and (need_g or need_b or need_w):
            sums = torch.empty((2, O), dtype=torch.float32, device=gz.device)
            _lib.check(L.kgdet_bn_fold_finish(_p(partial), max(P, 1), _p(weight), _p(gw), _p(s), _p(mean), _p(var), ctx.eps,
                                              _p(sums[0]), _p(sums[1]) if gw is not None else None, O,
                                              weight.numel() // O, st), 'bn_fold_finish')
        gres = g if (ctx.has_res and ctx.needs_input_grad[7]) else None
        return (gx, gw if need_w else None, sums[1] if need_g else None, sums[0] if need_b else None, None, None, None, gres,
                None, None, None, None, None)


FUSE_STEM = _os.environ.get('KGDET_FUSE_STEM', '1') == '1'   # 0 / False: conv_bn + nn.MaxPool2d (the tests compare the two)
MERGE_CONV_BN = True    # False: two nodes (conv_split, frozen_bn_act) -- the tests compare the two
SKIP_ALIAS = _os.environ.get('KGDET_SKIP_ALIAS', '1') == '1'   # identity-branch gradient added inside conv1's grad_input (0: A/B)


def _fused_bn_ok(x, bn, residual):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and not bn.training and bn.track_running_stats
            and (residual is None or (residual.dtype == x.dtype and residual.is_contiguous())))


def frozen_bn_act(x, bn, residual=None, relu=False):
    """BatchNorm with frozen statistics (eval mode) + residual add + ReLU.  On the GPU in fp32 this is the fused
    HIP op; otherwise (CPU tests, autocast dtypes, exotic layouts) the three torch ops of the reference."""
    if (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4 and not bn.training
            and bn.track_running_stats and x.shape[0] * x.shape[1] <= 65535
            and (residual is None or (residual.dtype == x.dtype and residual.shape == x.shape
                                      and residual.is_contiguous()))):
        return _FrozenBNAct.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, residual, relu)
    out = bn(x)
    if residual is not None:
        out = out + residual
    return F.relu(out, inplace=True) if relu else out


class _Subsample2(torch.autograd.Function):
    """``x[:, :, ::2, ::2]`` as a contiguous tensor; backward = the zero-stuffed gradient in ONE pass (csrc/glue.hip)
    instead of slice_backward's zero fill + strided copy"""

    @staticmethod
    def forward(ctx, x):
        from . import _lib
        B, C, H, W = x.shape
        y = x.new_empty(B, C, (H + 1) // 2, (W + 1) // 2)
        _lib.check(_lib.lib().kgdet_subsample2_forward(_lib.ptr(x), _lib.ptr(y), ctypes.c_int64(B * C), ctypes.c_int32(H),
                                                       ctypes.c_int32(W), _lib.current_stream()), 'subsample2_forward')
        ctx.shape = (B, C, H, W)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        from . import _lib
        B, C, H, W = ctx.shape
        gx = gy.new_empty(B, C, H, W)
        _lib.check(_lib.lib().kgdet_subsample2_backward(_lib.ptr(gy.contiguous()), None, _lib.ptr(gx), ctypes.c_int64(B * C),
                                                        ctypes.c_int32(H), ctypes.c_int32(W), _lib.current_stream()),
                   'subsample2_backward')
        return gx


def _subsample2(x):
    if x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.shape[3] % 4 == 0 and x.shape[2] % 2 == 0:
        return _Subsample2.apply(x)
    return x[:, :, ::2, ::2].contiguous()


FUSED_RESIDUAL_1X1 = _os.environ.get('KGDET_INFER_FUSED_RES', '1') == '1'   # conv3 + bn3 + add + ReLU as one kernel where measured faster (0: A/B)
GEMM_1X1 = _os.environ.get('KGDET_INFER_GEMM_1X1', '1') == '1'     # 0: every inference convolution through MIOpen (A/B)
_gemm_choice = {}       # (Cin, Cout, B, H, W, residual?, relu) -> True: hipBLASLt GEMM, False: MIOpen convolution


def _load_gemm_choices():
    """KGDET_GEMM_CHOICES=<file.json>: the measured route choices are read from the file when it exists (so that a PROFILED process
    takes the routes an un-profiled one measured -- a tracer's per-launch overhead shifts the timings the choice is made on) and
    written to it at exit otherwise."""
    path = _os.environ.get('KGDET_GEMM_CHOICES')
    if not path:
        return
    import atexit, json
    if _os.path.isfile(path):
        for k, v in json.load(open(path)):
            _gemm_choice[tuple(k)] = v
    else:
        atexit.register(lambda: json.dump([[list(k), v] for k, v in _gemm_choice.items()], open(path, 'w')))


_load_gemm_choices()


def fused_residual_ready(conv3, B, H, W):
    """True when conv3 + bn3 + add + ReLU of this shape runs on the fused kernel (measured choice): the caller may then hand it
    conv2's RAW output and bias (`_conv_bn(..., in_bias=...)`) instead of running conv2's epilogue pass"""
    return (FUSED_RESIDUAL_1X1 and GEMM_1X1 and
            _gemm_choice.get((conv3.in_channels, conv3.out_channels, B, H, W, True, True)) == 'fused')


_bias_sums = {}     # (id(bias), id(other)) -> (weakref(bias), weakref(other), bias + other)


def _summed_bias(bias, other):
    """bias + other (fp32 [C]), cached for as long as both tensors live (the folded biases of conv3 and of the downsample branch:
    entries of _fold_cache, replaced -- never updated in place -- when a weight changes)"""
    if other is None:
        return bias
    key = (id(bias), id(other))
    hit = _bias_sums.get(key)
    if hit is None or hit[0]() is not bias or hit[1]() is not other:
        if len(_bias_sums) > 256:      # (ids of dead tensors: drop what no longer resolves)
            for k in [k for k, v in _bias_sums.items() if v[0]() is None or v[1]() is None]:
                del _bias_sums[k]
        hit = _bias_sums[key] = (weakref.ref(bias), weakref.ref(other), (bias + other).contiguous())
    return hit[2]


def _conv1x1_as_gemm(conv, hit, x, residual, relu, in_bias=None, res_bias=None):
    """bf16 channels-last inference: a 1x1 stride-1 convolution IS the GEMM [B*H*W, Cin] x [Cin, Cout] on the channels-last
    storage, and hipBLASLt takes the folded-BatchNorm bias (+ ReLU) as its epilogue -- no separate bias / ReLU pass over the
    activation (csrc/epilogue.hip bias_act_nhwc was the largest kernel of the inference batch).  With a residual the GEMM adds
    it as its C matrix and only bias + ReLU remain as a pass.  MIOpen's NHWC kernels win on the large early maps, the GEMM on
    the deep layers (tools/bench_1x1_gemm.py: 1024 -> 256 at 50 x 84: 41 -> 24 us, 64 -> 64 at 200 x 336: 45 -> 86 us), so
    the choice is MEASURED once per shape during the eager warm-up calls (never while a graph is being captured); returns
    None when the convolution path should run."""
    B, cin, H, W = x.shape
    cout = hit[1].shape[0]
    key = (cin, cout, B, H, W, residual is not None, bool(relu))
    choice = _gemm_choice.get(key)
    if choice is False or (choice is None and torch.cuda.is_current_stream_capturing()):
        return None
    w2 = hit[1].view(cout, cin)
    bias16 = hit[5] if len(hit) > 5 else None
    # res_bias: the residual is the RAW output of the downsample convolution, whose folded bias joins this one's
    bias32 = _summed_bias(hit[2], res_bias)

    def gemm():
        x2 = x.permute(0, 2, 3, 1).reshape(-1, cin)
        if residual is not None:
            y2 = torch.addmm(residual.permute(0, 2, 3, 1).reshape(-1, cout), x2, w2.t())
            y = y2.view(B, H, W, cout).permute(0, 3, 1, 2)
            return _epilogue_(y, bias32, None, relu)
        if relu:
            y2 = torch._addmm_activation(bias16, x2, w2.t(), use_gelu=False)
        else:
            y2 = torch.addmm(bias16, x2, w2.t())
        return y2.view(B, H, W, cout).permute(0, 3, 1, 2)

    def fused():
        # conv3 + bn3 + identity + ReLU as ONE kernel (csrc/conv_nhwc.hip): x, the residual and the output cross the fabric once;
        # with in_bias, x is conv2's RAW output and its bias + ReLU happen as the activations are loaded
        # (residual None: conv1 + bn1 + ReLU, the same kernel without the identity add)
        from . import _lib
        y = torch.empty((B, cout, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        _lib.check(_lib.lib().kgdet_conv1x1_nhwc_residual_in(
            _lib.ptr(x), _lib.ptr(in_bias), _lib.ptr(hit[1]), _lib.ptr(bias32), _lib.ptr(residual), _lib.ptr(y),
            ctypes.c_int64(B * H * W), ctypes.c_int32(cin), ctypes.c_int32(cout), ctypes.c_int32(1 if relu else 0),
            _lib.current_stream()), 'conv1x1_nhwc_residual')
        return y

    ext = NHWC_EXTENDED      # (round 5: no residual, N = 64, K up to 512; 0 = round 4's envelope, A/B)
    fused_ok = (cin % 16 == 0 and cin <= (512 if ext else 384) and (cout % 128 == 0 or (ext and cout == 64))
                and (ext or residual is not None)
                and x.is_contiguous(memory_format=torch.channels_last) and hit[1].is_contiguous(memory_format=torch.channels_last)
                and (residual is None or (residual.is_contiguous(memory_format=torch.channels_last)
                                          and residual.shape == (B, cout, H, W))))
    if choice == 'fused' and fused_ok:   # the choice is keyed by shape; layout / contiguity are properties of THIS call
        return fused()
    if in_bias is not None:          # (only passed once the choice is 'fused', fused_residual_ready: the caller's problem)
        return None
    if choice == 'fused':            # same shape, other layout: the GEMM path below handles any strides
        return gemm()
    if choice is None:
        def conv_path():
            y = F.conv2d(x, hit[1], None, conv.stride, conv.padding, conv.dilation, conv.groups)
            return _epilogue_(y, bias32, residual, relu)

        def timed(fn):
            for _ in range(2):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1)
        t_conv, t_gemm = timed(conv_path), timed(gemm)
        t_fused = timed(fused) if (fused_ok and FUSED_RESIDUAL_1X1) else float('inf')
        if t_fused < 0.95 * min(t_conv, t_gemm):
            choice = 'fused'
        else:
            choice = t_gemm < 0.95 * t_conv
        _gemm_choice[key] = choice
        if choice == 'fused':
            return fused()
        if not choice:
            return None
    return gemm()


def conv_bn(conv, bn, x, relu=False, residual=None, skip=False, gate_in=None, gate_out=None):
    """``[relu](bn(conv(x)) [+ residual])``; ``skip=True``: returns (that, x) where the second is x or an alias of it whose
    gradient is folded into this convolution's grad_input (_ConvBNAct).  ``gate_in`` / ``gate_out``: _GateLink objects of the
    caller, who vouches that x has no other consumer (gate_out) -- see _GateLink; honoured by the folded training node only."""
    out = _conv_bn(conv, bn, x, relu, residual, skip, gate_in=gate_in, gate_out=gate_out)
    if skip and not isinstance(out, tuple):
        return out, x
    return out


def _conv_bn(conv, bn, x, relu=False, residual=None, skip=False, raw=False, in_bias=None, gate_in=None, gate_out=None,
             res_bias=None):
    """conv_bn's body (with ``skip`` the training fast path may return the (output, alias) pair itself).  In inference (autograd off, BatchNorm in eval mode, plain bias-free
    Conv2d) the frozen statistics are folded into the convolution -- w' = w * gamma / sigma,
    b' = beta - mu * gamma / sigma -- and bias, residual add and ReLU run as ONE in-place pass over the activation
    instead of BatchNorm + add + clamp (three).  Under bf16 autocast the folded weight is kept in bf16 and
    channels-last: MIOpen's bf16 kernels on gfx950 are NHWC implicit GEMMs, so NCHW activations cost a transpose
    in and out of every convolution (6 % of the batch).  The folded tensors are cached per conv and dropped
    whenever the backbone changes mode (``ResNet.train``)."""
    if (bn.training or not isinstance(bn, _BatchNorm) or type(conv) is not nn.Conv2d or conv.bias is not None
            or not bn.track_running_stats):
        out = bn(conv(x))
        if residual is not None:
            out += residual
        return F.relu(out, inplace=True) if relu else out
    if torch.is_grad_enabled():
        if conv1x1.applicable(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups):
            if (MERGE_CONV_BN and _fused_bn_ok(x, bn, residual) and x.shape[0] * conv.weight.shape[0] <= 65535
                    and (residual is None or residual.shape[1] == conv.weight.shape[0])):
                fold = conv1x1.fold_images(conv.weight, bn) if bn.affine else None
                if fold is not None:     # from the pair's second step on: BatchNorm folded into the convolution
                    if skip and SKIP_ALIAS and x.requires_grad:
                        return _ConvBNActFold.apply(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                                    bn.eps, residual, relu, True, fold, gate_in, gate_out)
                    # (skip without the alias: x has a second consumer, the identity branch -- its gradient must not be gated)
                    out = _ConvBNActFold.apply(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                                               residual, relu, False, fold, gate_in, None if skip else gate_out)
                    return (out, x) if skip else out
                if skip and SKIP_ALIAS and x.requires_grad:
                    return _ConvBNAct.apply(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                                            residual, relu, True)
                out = _ConvBNAct.apply(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                                       residual, relu)
                return (out, x) if skip else out
            return frozen_bn_act(conv1x1.conv_split(x, conv.weight), bn, residual, relu)   # split-bf16 MFMA GEMMs
        if conv1x1.applicable_stride2(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups):
            return frozen_bn_act(conv1x1.conv3x3_stride2(x, conv.weight), bn, residual, relu)
        if (conv.kernel_size == (1, 1) and conv.stride == (2, 2) and conv.padding == (0, 0) and x.is_cuda
                and x.dtype == torch.float32):
            # stride-2 1x1 (the downsample branch): a 1x1 convolution of the subsampled input.  MIOpen's fp32 strided
            # kernels run at 13-18 TFLOP/s backward; the quarter-size copy + the split-bf16 GEMMs are ~2x faster
            xs = _subsample2(x)
            if conv1x1.applicable(xs, conv.weight):
                fold = (conv1x1.fold_images(conv.weight, bn)
                        if (MERGE_CONV_BN and bn.affine and _fused_bn_ok(xs, bn, residual)
                            and xs.shape[0] * conv.weight.shape[0] <= 65535) else None)
                if fold is not None:
                    return _ConvBNActFold.apply(xs, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                                                residual, relu, False, fold)
                return frozen_bn_act(conv1x1.conv_split(xs, conv.weight), bn, residual, relu)
        return frozen_bn_act(conv(x), bn, residual, relu)
    bf16 = x.is_cuda and (x.dtype == torch.bfloat16 or (torch.is_autocast_enabled()
                                                        and torch.get_autocast_dtype('cuda') == torch.bfloat16))
    hit = _fold_cache.get((id(conv), bf16))
    # staleness: in-place updates (load_state_dict's copy_, optimizer steps that bump the counter) change _version
    ver = (conv.weight._version, bn.running_mean._version, bn.running_var._version,
           bn.weight._version if bn.affine else 0, bn.bias._version if bn.affine else 0)
    if hit is None or hit[0]() is not conv or hit[4] != ver:
        scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps) if bn.affine else torch.rsqrt(bn.running_var + bn.eps)
        shift = (bn.bias if bn.affine else 0) - bn.running_mean * scale
        w = (conv.weight * scale.view(-1, 1, 1, 1)).detach()
        if bf16:
            w = w.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        packed = None
        if not bf16 and conv1x1.applicable(x, w, conv.stride, conv.padding, conv.dilation, conv.groups):
            packed = conv1x1._pack(w.contiguous(), False)     # fp32 inference: split-bf16 MFMA kernels, packed once
        hit = (weakref.ref(conv), w, shift.detach().float().contiguous(), packed, ver,
               shift.detach().to(torch.bfloat16).contiguous() if bf16 else None)
        _fold_cache[(id(conv), bf16)] = hit
    if bf16 and not x.is_contiguous(memory_format=torch.channels_last):
        x = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    if (bf16 and GEMM_1X1 and not raw and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0)
            and conv.groups == 1 and x.dtype == torch.bfloat16 and (residual is None or residual.dtype == torch.bfloat16)):
        out = _conv1x1_as_gemm(conv, hit, x, residual, relu, in_bias, res_bias)
        if out is not None:
            return out
    if in_bias is not None:      # the convolution route after all: conv2's epilogue as its own pass
        x = _epilogue_(x, in_bias, None, True)
    bias = _summed_bias(hit[2], res_bias)     # (res_bias: `residual` is a RAW convolution output whose bias is added here)
    if hit[3] is not None and x.dtype == torch.float32 and x.is_contiguous() and x.shape[2] * x.shape[3] % 2 == 0:
        if _FUSE_EPI and (residual is None or (residual.dtype == torch.float32 and residual.is_contiguous())):
            # bias, residual and ReLU ride on the convolution's store: no separate epilogue pass
            return conv1x1._apply(hit[3], x, hit[1].shape[0], hit[1].shape[2] * hit[1].shape[3], 1, bias, residual, relu)
        out = conv1x1._apply(hit[3], x, hit[1].shape[0], hit[1].shape[2] * hit[1].shape[3])
    else:
        out = F.conv2d(x, hit[1], None, conv.stride, conv.padding, conv.dilation, conv.groups)
    if raw:     # (the stem: the caller fuses bias + ReLU with the pooling)
        return out, hit[2]
    return _epilogue_(out, bias, residual, relu)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style='pytorch', with_cp=False,
                 conv_cfg=None, norm_cfg=dict(type='BN'), dcn=None, gcb=None, gen_attention=None):
        super(Bottleneck, self).__init__()
        assert style in ['pytorch', 'caffe']
        assert dcn is None or isinstance(dcn, dict)
        assert gcb is None and gen_attention is None, 'context / attention plugins are outside the KGDet path'
        self.inplanes = inplanes
        self.planes = planes
        self.stride = stride
        self.dilation = dilation
        self.style = style
        self.with_cp = with_cp
        self.conv_cfg = conv_cfg
        self.norm_cfg = norm_cfg
        self.dcn = dcn
        self.with_dcn = dcn is not None
        # "pytorch": the 3x3 conv carries the stride; "caffe": the first 1x1 does
        self.conv1_stride, self.conv2_stride = (1, stride) if style == 'pytorch' else (stride, 1)

        self.norm1_name, norm1 = build_norm_layer(norm_cfg, planes, postfix=1)
        self.norm2_name, norm2 = build_norm_layer(norm_cfg, planes, postfix=2)
        self.norm3_name, norm3 = build_norm_layer(norm_cfg, planes * self.expansion, postfix=3)

        self.conv1 = build_conv_layer(conv_cfg, inplanes, planes, kernel_size=1, stride=self.conv1_stride, bias=False)
        self.add_module(self.norm1_name, norm1)
        fallback_on_stride = False
        self.with_modulated_dcn = False
        if self.with_dcn:
            fallback_on_stride = dcn.get('fallback_on_stride', False)
            self.with_modulated_dcn = dcn.get('modulated', False)
        if not self.with_dcn or fallback_on_stride:
            self.conv2 = build_conv_layer(conv_cfg, planes, planes, kernel_size=3, stride=self.conv2_stride,
                                          padding=dilation, dilation=dilation, bias=False)
        else:
            assert conv_cfg is None, 'conv_cfg must be None for DCN'
            deformable_groups = dcn.get('deformable_groups', 1)
            conv_op, offset_channels = ((dcn_ops.ModulatedDeformConv, 27) if self.with_modulated_dcn else
                                        (dcn_ops.DeformConv, 18))
            self.conv2_offset = nn.Conv2d(planes, deformable_groups * offset_channels, kernel_size=3,
                                          stride=self.conv2_stride, padding=dilation, dilation=dilation)
            self.conv2 = conv_op(planes, planes, kernel_size=3, stride=self.conv2_stride, padding=dilation,
                                 dilation=dilation, deformable_groups=deformable_groups, bias=False)
        self.add_module(self.norm2_name, norm2)
        self.conv3 = build_conv_layer(conv_cfg, planes, planes * self.expansion, kernel_size=1, bias=False)
        self.add_module(self.norm3_name, norm3)

        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    @property
    def norm1(self):
        return getattr(self, self.norm1_name)

    @property
    def norm2(self):
        return getattr(self, self.norm2_name)

    @property
    def norm3(self):
        return getattr(self, self.norm3_name)

    def forward(self, x):
        identity = x
        # gradient hand-overs (_GateLink): l1 / l2 inside the block; `arriving` = the link the previous block of this layer left
        # on its output (x is then that block's relu(.) and this block -- conv1 with the skip alias -- its only consumer);
        # `leaving` goes out on this block's output for the next block.  Training on the folded path only (see conv_bn).
        fuse = GATE_FUSION and torch.is_grad_enabled() and not self.with_dcn
        l1, l2, leaving = (_GateLink(), _GateLink(), _GateLink()) if fuse else (None, None, None)
        arriving = x.__dict__.pop('_kgdet_gate_link', None) if fuse else None
        if self.downsample is None and not self.with_dcn:
            out, identity = conv_bn(self.conv1, self.norm1, x, relu=True, skip=True, gate_in=l1, gate_out=arriving)
        else:
            out = conv_bn(self.conv1, self.norm1, x, relu=True, gate_in=l1)
        if not self.with_dcn:
            infer16 = not torch.is_grad_enabled() and out.dtype == torch.bfloat16 and RAW_BRANCHES
            shift_ds = None
            if (self.downsample is not None and infer16 and isinstance(self.downsample, nn.Sequential) and len(self.downsample) == 2
                    and getattr(self.downsample[0], 'stride', None) == (2, 2)):     # (stride 1: the GEMM route has the bias in its epilogue)
                # bf16 inference: the downsample branch stays RAW (no bias pass over the widest activation of the block); its
                # folded bias joins conv3's, whose epilogue adds the residual anyway -- one rounding to bf16 fewer
                res = _conv_bn(self.downsample[0], self.downsample[1], x, raw=True)
                if isinstance(res, tuple):
                    identity, shift_ds = res
                else:
                    identity = res
            s2 = self.conv2.stride[0]
            if (infer16 and self.conv2.stride in ((1, 1), (2, 2))
                    and fused_residual_ready(self.conv3, out.shape[0], (out.shape[2] + s2 - 1) // s2, (out.shape[3] + s2 - 1) // s2)):
                # bf16 inference: conv2's bias + ReLU ride on the activation loads of the fused conv3 kernel
                out, shift2 = _conv_bn(self.conv2, self.norm2, out, relu=True, raw=True)
                if self.downsample is not None and shift_ds is None and identity is x:
                    identity = conv_bn(self.downsample[0], self.downsample[1], x)
                return _conv_bn(self.conv3, self.norm3, out, relu=True, residual=identity, in_bias=shift2, res_bias=shift_ds)
            out = conv_bn(self.conv2, self.norm2, out, relu=True, gate_in=l2, gate_out=l1)
            if self.downsample is not None and shift_ds is None and identity is x:
                identity = conv_bn(self.downsample[0], self.downsample[1], x)
            if shift_ds is not None:
                return _conv_bn(self.conv3, self.norm3, out, relu=True, residual=identity, res_bias=shift_ds)
            out = conv_bn(self.conv3, self.norm3, out, relu=True, residual=identity, gate_in=leaving, gate_out=l2)
            if fuse:
                out._kgdet_gate_link = leaving      # (taken -- and removed -- by the next bottleneck of the layer, if there is one)
            return out
        elif self.with_modulated_dcn:
            offset_mask = self.conv2_offset(out)
            offset = offset_mask[:, :18, :, :]
            mask = offset_mask[:, -9:, :, :].sigmoid()
            out = self.conv2(out, offset, mask)
        else:
            out = self.conv2(out, self.conv2_offset(out))
        out = self.relu(self.norm2(out))
        out = self.norm3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out += identity
        return self.relu(out)


def make_res_layer(block, inplanes, planes, blocks, stride=1, dilation=1, style='pytorch', with_cp=False,
                   conv_cfg=None, norm_cfg=dict(type='BN'), dcn=None, gcb=None, gen_attention=None,
                   gen_attention_blocks=[]):
    downsample = None
    if stride != 1 or inplanes != planes * block.expansion:
        downsample = nn.Sequential(
            build_conv_layer(conv_cfg, inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
            build_norm_layer(norm_cfg, planes * block.expansion)[1],
        )
    layers = [block(inplanes=inplanes, planes=planes, stride=stride, dilation=dilation, downsample=downsample,
                    style=style, with_cp=with_cp, conv_cfg=conv_cfg, norm_cfg=norm_cfg, dcn=dcn, gcb=gcb,
                    gen_attention=None)]
    inplanes = planes * block.expansion
    for _ in range(1, blocks):
        layers.append(block(inplanes=inplanes, planes=planes, stride=1, dilation=dilation, style=style,
                            with_cp=with_cp, conv_cfg=conv_cfg, norm_cfg=norm_cfg, dcn=dcn, gcb=gcb,
                            gen_attention=None))
    return nn.Sequential(*layers)


@BACKBONES.register_module
class ResNet(nn.Module):
    arch_settings = {
        18: (BasicBlock, (2, 2, 2, 2)),
        34: (BasicBlock, (3, 4, 6, 3)),
        50: (Bottleneck, (3, 4, 6, 3)),
        101: (Bottleneck, (3, 4, 23, 3)),
        152: (Bottleneck, (3, 8, 36, 3))
    }

    def __init__(self, depth, num_stages=4, strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3),
                 style='pytorch', frozen_stages=-1, conv_cfg=None, norm_cfg=dict(type='BN', requires_grad=True),
                 norm_eval=True, dcn=None, stage_with_dcn=(False, False, False, False), gcb=None,
                 stage_with_gcb=(False, False, False, False), gen_attention=None,
                 stage_with_gen_attention=((), (), (), ()), with_cp=False, zero_init_residual=True):
        super(ResNet, self).__init__()
        if depth not in self.arch_settings:
            raise KeyError('invalid depth {} for resnet'.format(depth))
        assert gcb is None and gen_attention is None, 'context / attention plugins are outside the KGDet path'
        assert 1 <= num_stages <= 4 and len(strides) == len(dilations) == num_stages and max(out_indices) < num_stages
        assert dcn is None or len(stage_with_dcn) == num_stages
        # (the attribute names are the reference's: resnet.py:386-404; configs and checkpoints address them)
        for key, value in dict(depth=depth, num_stages=num_stages, strides=strides, dilations=dilations,
                               out_indices=out_indices, style=style, frozen_stages=frozen_stages, conv_cfg=conv_cfg,
                               norm_cfg=norm_cfg, with_cp=with_cp, norm_eval=norm_eval, dcn=dcn,
                               stage_with_dcn=stage_with_dcn, zero_init_residual=zero_init_residual).items():
            setattr(self, key, value)
        self.block, per_stage = self.arch_settings[depth]
        self.stage_blocks = per_stage[:num_stages]
        self.inplanes = 64
        self._make_stem_layer()
        self.res_layers = []
        for stage, num_blocks in enumerate(self.stage_blocks):
            planes = 64 << stage
            layer = make_res_layer(self.block, self.inplanes, planes, num_blocks, stride=strides[stage],
                                   dilation=dilations[stage], style=style, with_cp=with_cp, conv_cfg=conv_cfg,
                                   norm_cfg=norm_cfg, dcn=dcn if stage_with_dcn[stage] else None)
            self.inplanes = planes * self.block.expansion
            self.res_layers.append('layer%d' % (stage + 1))
            self.add_module(self.res_layers[-1], layer)
        self._freeze_stages()
        self.feat_dim = self.block.expansion * (64 << (len(self.stage_blocks) - 1))

    @property
    def norm1(self):
        return getattr(self, self.norm1_name)

    def _make_stem_layer(self):
        self.conv1 = build_conv_layer(self.conv_cfg, 3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.norm1_name, norm1 = build_norm_layer(self.norm_cfg, 64, postfix=1)
        self.add_module(self.norm1_name, norm1)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)

    def _freeze_stages(self):
        """stem (frozen_stages >= 0) and layer1 .. layer<frozen_stages>: eval mode, no gradients (resnet.py:466-478)"""
        frozen = [self.conv1, self.norm1] if self.frozen_stages >= 0 else []
        frozen += [getattr(self, 'layer%d' % i) for i in range(1, self.frozen_stages + 1)]
        for module in frozen:
            if module is not self.conv1:
                module.eval()
            for p in module.parameters():
                p.requires_grad = False

    def init_weights(self, pretrained=None):
        if isinstance(pretrained, str):
            from .checkpoint import load_checkpoint
            load_checkpoint(self, pretrained, strict=False)
        elif pretrained is None:
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    kaiming_init(m)
                elif isinstance(m, (_BatchNorm, nn.GroupNorm)):
                    constant_init(m, 1)
            if self.dcn is not None:
                for m in self.modules():
                    if isinstance(m, Bottleneck) and hasattr(m, 'conv2_offset'):
                        constant_init(m.conv2_offset, 0)
            if self.zero_init_residual:
                for m in self.modules():
                    if isinstance(m, Bottleneck):
                        constant_init(m.norm3, 0)
                    elif isinstance(m, BasicBlock):
                        constant_init(m.norm2, 0)
        else:
            raise TypeError('pretrained must be a str or None')

    def _stem(self, x):
        """maxpool(relu(norm1(conv1(x)))) (resnet.py:528).  A frozen stem in fp32 on the GPU: BatchNorm, ReLU and the
        pooling are one pass over conv1's output (csrc/bn_act.hip bn_relu_maxpool)"""
        bn, mp = self.norm1, self.maxpool
        if (FUSE_STEM and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled()
                and isinstance(bn, _BatchNorm) and not bn.training and bn.track_running_stats and type(self.conv1) is nn.Conv2d
                and not (torch.is_grad_enabled() and (x.requires_grad or self.conv1.weight.requires_grad
                                                      or (bn.affine and (bn.weight.requires_grad or bn.bias.requires_grad))))
                and (mp.kernel_size, mp.stride, mp.padding, mp.dilation, mp.ceil_mode) == (3, 2, 1, 1, False)
                and x.shape[0] * self.conv1.out_channels <= 65535):
            from . import _lib
            with torch.no_grad():
                y = _stem_conv(self.conv1, x)
                N, C, H, W = y.shape
                out = torch.empty((N, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=y.dtype, device=y.device)
                _lib.check(_bn_lib().kgdet_bn_relu_maxpool(
                    _p(y), _p(bn.weight), _p(bn.bias), _p(bn.running_mean), _p(bn.running_var), bn.eps, _p(out), N, C, H, W,
                    _lib.raw_stream(y.device.index)), 'bn_relu_maxpool')
            return out
        if (FUSE_STEM and x.is_cuda and not torch.is_grad_enabled() and isinstance(bn, _BatchNorm) and not bn.training
                and bn.track_running_stats and type(self.conv1) is nn.Conv2d and self.conv1.bias is None
                and (mp.kernel_size, mp.stride, mp.padding, mp.dilation, mp.ceil_mode) == (3, 2, 1, 1, False)):
            # inference, channels-last (bf16 autocast): BatchNorm is folded into conv1; bias + ReLU + pooling as one pass
            y, shift = _conv_bn(self.conv1, bn, x, relu=True, raw=True)
            C = y.shape[1]
            if (y.dtype in (torch.float32, torch.bfloat16) and y.dim() == 4 and not y.is_contiguous()
                    and y.is_contiguous(memory_format=torch.channels_last) and C % (4 if y.dtype == torch.float32 else 8) == 0):
                from . import _lib
                N, _, H, W = y.shape
                out = torch.empty((N, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=y.dtype, device=y.device,
                                  memory_format=torch.channels_last)
                _lib.check(_lib.lib().kgdet_bias_relu_maxpool_nhwc(
                    _lib.ptr(y), _lib.ptr(shift), _lib.ptr(out), ctypes.c_int64(N), ctypes.c_int32(C), ctypes.c_int32(H),
                    ctypes.c_int32(W), ctypes.c_int32(0 if y.dtype == torch.float32 else 1), _lib.current_stream()),
                    'bias_relu_maxpool_nhwc')
                return out
            return self.maxpool(_epilogue_(y, shift, None, True))
        return self.maxpool(conv_bn(self.conv1, self.norm1, x, relu=True))

    def forward(self, x):
        x = self._stem(x)
        outs = []
        for i, layer_name in enumerate(self.res_layers):
            x = getattr(self, layer_name)(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def train(self, mode=True):
        super(ResNet, self).train(mode)
        clear_fold_cache()   # weights may have changed since the last inference pass
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, _BatchNorm):
                    m.eval()
