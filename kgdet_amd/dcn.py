"""Deformable convolution v1 / v2 modules and functions -- host-side mirror of the reference's
``mmdet/ops/dcn/deform_conv.py`` (same class names, constructor arguments, parameter names and
error behaviour) on top of the HIP kernels in libkgdet_hip.so.

Reference interface mirrored (R = mmdetection/mmdet/ops/dcn/deform_conv.py):
  DeformConvFunction :12-110      deform_conv :186           DeformConv :190-236
  DeformConvPack :239-261         ModulatedDeformConvFunction :113-183
  ModulatedDeformConv :264-304    ModulatedDeformConvPack :307-337

As in the reference there is no CPU implementation: non-GPU tensors raise NotImplementedError.
"""
import ctypes
import os
import math
import weakref

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from . import _lib

_workspaces = {}


def _workspace(device, nbytes):
    """Persistent per-(device, stream) scratch for split-K partial tiles (caller-owned memory in
    the C ABI; the reference re-allocates its column buffer with at::zeros on every call)."""
    key = (device.index, _lib.raw_stream(device.index))
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def _shape(input, weight, stride, padding, dilation, groups, deformable_groups):
    s = _lib.DcnShape()
    s.N, s.C, s.H, s.W = input.shape
    s.O, _, s.kh, s.kw = weight.shape
    s.stride_h, s.stride_w = stride
    s.pad_h, s.pad_w = padding
    s.dil_h, s.dil_w = dilation
    s.groups = groups
    s.deformable_groups = deformable_groups
    return s


def _output_size(input, weight, padding, dilation, stride):
    """R:96-110"""
    channels = weight.size(0)
    output_size = (input.size(0), channels)
    for d in range(input.dim() - 2):
        in_size = input.size(d + 2)
        pad = padding[d]
        kernel = dilation[d] * (weight.size(d + 2) - 1) + 1
        stride_ = stride[d]
        output_size += ((in_size + (2 * pad) - kernel) // stride_ + 1, )
    if not all(map(lambda s: s > 0, output_size)):
        raise ValueError('convolution input is too small (output would be {})'.format(
            'x'.join(map(str, output_size))))
    return output_size


# Forward arithmetic of the HIP kernels (include/kgdet_hip.h, KGDET_DCN_*):
#   'split' (default)  bf16 MFMA on a hi/lo split of both fp32 operands: fp32-accurate (~1e-6 of the output scale)
#   'bf16'             operands rounded to bf16 once, fp32 accumulate -- what autocast(bfloat16) inference asks for
#   'exact'            v_mfma_f32_32x32x2_f32, bit-exact fp32 products
_FORWARD_PRECISION = 'split'
_PRECISION_FLAGS = {'split': 0, 'bf16': _lib.DCN_BF16, 'exact': _lib.DCN_EXACT_FP32}


def set_forward_precision(mode):
    """Select the forward arithmetic ('split' | 'bf16' | 'exact'); returns the previous mode."""
    global _FORWARD_PRECISION
    if mode not in _PRECISION_FLAGS:
        raise ValueError('unknown DeformConv forward precision {!r}'.format(mode))
    prev, _FORWARD_PRECISION = _FORWARD_PRECISION, mode
    return prev


class forward_precision(object):
    """Context manager: ``with dcn.forward_precision('bf16'): ...``"""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = set_forward_precision(self.mode)

    def __exit__(self, *exc):
        set_forward_precision(self.prev)


_EXACT_BACKWARD = False


class arithmetic(object):
    """``with dcn.arithmetic('exact'):`` -- the whole step in plain fp32 arithmetic: deformable forward on the
    f32-input MFMA kernel, deformable backward on the exact-fp32 kernels (instead of the split-bf16 plane kernels) and
    the dense convolutions on MIOpen's fp32 kernels (instead of csrc/conv1x1.hip).  ``'split'`` is the default
    everywhere.  Used to measure what the hi/lo split costs in accuracy over a whole training step."""

    def __init__(self, mode):
        assert mode in ('split', 'exact')
        self.mode = mode

    def __enter__(self):
        global _EXACT_BACKWARD
        from . import conv1x1
        self.prev = (set_forward_precision(self.mode), _EXACT_BACKWARD, conv1x1.ENABLED)
        _EXACT_BACKWARD = self.mode == 'exact'
        conv1x1.ENABLED = self.mode != 'exact'
        _lib.check(_lib.lib().kgdet_set_option(0, int(_EXACT_BACKWARD)), 'kgdet_set_option')

    def __exit__(self, *exc):
        global _EXACT_BACKWARD
        from . import conv1x1
        set_forward_precision(self.prev[0])
        _EXACT_BACKWARD, conv1x1.ENABLED = self.prev[1], self.prev[2]
        _lib.check(_lib.lib().kgdet_set_option(0, int(_EXACT_BACKWARD)), 'kgdet_set_option')


def _fwd_flags(relu):
    return ctypes.c_uint32((_lib.DCN_RELU if relu else 0) | _PRECISION_FLAGS[_FORWARD_PRECISION])


def _require_f32(*tensors):
    """the C ABI is float32 only; never let another dtype be read as float bits"""
    for t in tensors:
        if t is not None and t.dtype != torch.float32:
            raise TypeError('kgdet_amd deformable convolution expects float32 tensors, got {} '
                            '(under torch.autocast the public functions cast for you)'.format(t.dtype))


def _autocast_apply(apply, args):
    """torch.autocast contract of the deformable ops: like the reference's fp32-only CUDA op they take and
    return float32 (lower-precision activations are cast up on entry); under bfloat16 autocast the
    forward products use bf16 operands with fp32 accumulation (KGDET_DCN_BF16) unless a precision was
    chosen explicitly with set_forward_precision()."""
    global _inference_depth
    if not torch.is_grad_enabled():   # seen here, outside the autograd Function (inside it grad mode is always off)
        _inference_depth += 1
        try:
            return _autocast_apply_inner(apply, args)
        finally:
            _inference_depth -= 1
    return _autocast_apply_inner(apply, args)


def _autocast_apply_inner(apply, args):
    if not torch.is_autocast_enabled('cuda'):
        return apply(*args)
    cast = [a.float() if torch.is_tensor(a) and a.is_floating_point() and a.dtype != torch.float32 else a
            for a in args]
    bf16 = torch.get_autocast_dtype('cuda') == torch.bfloat16 and _FORWARD_PRECISION == 'split'
    with torch.autocast('cuda', enabled=False), forward_precision('bf16' if bf16 else _FORWARD_PRECISION):
        return apply(*cast)


_pack_cache = {}          # id(weight) -> (weakref, key, packed)
_PACK_CACHE_MAX = 64


def clear_pack_cache():
    """Drop cached weight images (only needed after mutating a weight through ``.data``, which bypasses
    torch's version counter)."""
    _pack_cache.clear()


_inference_depth = 0   # > 0 while a public op was entered with autograd disabled (torch.no_grad inference)


SPLIT_ONLY_PACKS = os.environ.get('KGDET_DCN_SPLIT_ONLY_PACKS', '1') == '1'   # 0: training packs all four images (A/B)


def split_images_suffice(shapes):
    """True when no product of these convolutions will read the fp32 weight images: the default arithmetic is selected and
    every shape has split-operand kernels for forward, grad_input, grad_offset and grad_weight"""
    if not SPLIT_ONLY_PACKS or _FORWARD_PRECISION == 'exact' or _EXACT_BACKWARD:
        return False
    L = _lib.lib()
    return all(L.kgdet_dcn_split_path_complete(ctypes.byref(s)) == 1 for s in shapes)


def pack_weights(weights, shapes, split_only=False):
    """pack_weight for several weights: those that need packing (all of them in training) go to the GPU as ONE launch
    (kgdet_dcn_pack_weight_images).  ``split_only``: only the bf16 hi/lo images (training: half the bytes written -- the caller
    vouches with ``split_images_suffice`` that nothing will read the fp32 images); cached inference packs are always complete."""
    cacheable = _inference_depth > 0
    split_only = bool(split_only) and not cacheable
    L = _lib.lib()
    out, todo = [None] * len(weights), []
    for i, (w, s) in enumerate(zip(weights, shapes)):
        nbytes = L.kgdet_dcn_packed_weight_bytes(ctypes.byref(s))
        key = (w.data_ptr(), w._version, s.groups, s.deformable_groups, tuple(w.shape), nbytes)
        if cacheable:
            hit = _pack_cache.get(id(w))
            if hit is not None and hit[0]() is w and hit[1] == key:
                out[i] = hit[2]
                continue
        out[i] = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
        todo.append((i, key))
    if todo:
        n = len(todo)
        shape_arr = (ctypes.POINTER(_lib.DcnShape) * n)(*[ctypes.pointer(shapes[i]) for i, _ in todo])
        w_arr = (ctypes.c_void_p * n)(*[weights[i].data_ptr() for i, _ in todo])
        p_arr = (ctypes.c_void_p * n)(*[out[i].data_ptr() for i, _ in todo])
        _lib.check(L.kgdet_dcn_pack_weight_images(ctypes.c_int32(n), shape_arr, w_arr, p_arr, ctypes.c_uint32(2 if split_only else 3),
                                                  _lib.current_stream()), 'kgdet_dcn_pack_weight_images')
        if cacheable:
            for i, key in todo:
                if len(_pack_cache) >= _PACK_CACHE_MAX:
                    _pack_cache.clear()
                _pack_cache[id(weights[i])] = (weakref.ref(weights[i]), key, out[i])
    return out


def pack_weight(weight, shape):
    """weight [O, C/g, kh, kw] -> the kernels' weight images (include/kgdet_hip.h, kgdet_dcn_pack_weight).
    Training packs on every call (its weights change every step, and fused optimizers update them WITHOUT
    bumping the tensor's ``_version``, so no cheap staleness check exists).  Inference under ``torch.no_grad()``
    reuses the images of an unchanged tensor -- same live object, storage and ``_version``; the cache is dropped
    whenever a DeformConv module changes mode (``model.train()`` / ``model.eval()``) and by clear_pack_cache()."""
    cacheable = _inference_depth > 0
    L = _lib.lib()
    nbytes = L.kgdet_dcn_packed_weight_bytes(ctypes.byref(shape))
    # (the packed images depend on the weight alone, never on the map size or the batch: one pack serves every pyramid
    # level -- csrc/dcn_api.hip kgdet_dcn_pack_weight)
    key = (weight.data_ptr(), weight._version, shape.groups, shape.deformable_groups, tuple(weight.shape), nbytes)
    if cacheable:
        hit = _pack_cache.get(id(weight))
        if hit is not None and hit[0]() is weight and hit[1] == key:
            return hit[2]
    packed = torch.empty(nbytes // 4, dtype=torch.float32, device=weight.device)
    _lib.check(L.kgdet_dcn_pack_weight(ctypes.byref(shape), _lib.ptr(weight), _lib.ptr(packed),
                                       _lib.current_stream()), 'kgdet_dcn_pack_weight')
    if cacheable:
        if len(_pack_cache) >= _PACK_CACHE_MAX:
            _pack_cache.clear()
        _pack_cache[id(weight)] = (weakref.ref(weight), key, packed)
    return packed


def _check_offset(offset, shape, out_size, mask=None):
    """The reference's shape_check messages (R/src/deform_conv_cuda.cpp:128-135) as RuntimeError."""
    K = shape.kh * shape.kw
    if offset.size(0) != shape.N:
        raise RuntimeError('invalid batch size of offset')
    if offset.size(2) != out_size[2] or offset.size(3) != out_size[3]:
        raise RuntimeError('invalid spatial size of offset, expected height: {} width: {}, but got '
                           'height: {} width: {}'.format(out_size[2], out_size[3], offset.size(2),
                                                         offset.size(3)))
    if offset.size(1) != shape.deformable_groups * 2 * K:
        raise RuntimeError('invalid number of channels of offset')
    if mask is not None and tuple(mask.shape) != (shape.N, shape.deformable_groups * K, out_size[2],
                                                   out_size[3]):
        raise RuntimeError('invalid shape of mask')


def _forward(input, offset, mask, weight, bias, shape, packed=None, relu=False):
    L = _lib.lib()
    _require_f32(input, offset, mask, weight, bias)
    input = input.contiguous()
    offset = offset.contiguous()
    mask = mask.contiguous() if mask is not None else None
    if packed is None:
        packed = pack_weight(weight.contiguous(), shape)
    out_size = _output_size(input, weight, (shape.pad_h, shape.pad_w), (shape.dil_h, shape.dil_w),
                            (shape.stride_h, shape.stride_w))
    _check_offset(offset, shape, out_size, mask)
    output = input.new_empty(out_size)
    ws_bytes = L.kgdet_dcn_workspace_bytes(ctypes.byref(shape))
    ws = _workspace(input.device, ws_bytes)
    _lib.check(L.kgdet_deform_conv_forward(
        ctypes.byref(shape), _lib.ptr(input), _lib.ptr(offset), _lib.ptr(mask), _lib.ptr(packed),
        _lib.ptr(bias), _lib.ptr(output), _fwd_flags(relu),
        _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.current_stream()), 'kgdet_deform_conv_forward')
    return output, packed


def grad_input_plane(input_size, offset, mask, weight, grad_output, shape, packed=None, bf16=False):
    """grad_input of a deformable conv on the plane kernel (kgdet_deform_conv_grad_input): transposed sampling,
    no atomics, deterministic.  ``input_size`` = (N, C, H, W)."""
    L = _lib.lib()
    _require_f32(offset, mask, weight, grad_output)
    if packed is None:
        packed = pack_weight(weight.contiguous(), shape)
    grad_output = grad_output.contiguous()
    grad_input = grad_output.new_empty(tuple(input_size))
    ws = _workspace(grad_output.device, L.kgdet_dcn_workspace_bytes(ctypes.byref(shape)))
    _lib.check(L.kgdet_deform_conv_grad_input(
        ctypes.byref(shape), _lib.ptr(offset.contiguous()), _lib.ptr(mask), _lib.ptr(packed), _lib.ptr(grad_output),
        _lib.ptr(grad_input), ctypes.c_uint32(_lib.DCN_BF16 if bf16 else 0), _lib.ptr(ws),
        ctypes.c_size_t(ws.numel()), _lib.current_stream()), 'kgdet_deform_conv_grad_input')
    return grad_input


def grad_offset_plane(input, offset, weight, grad_output, shape, packed=None, bf16=False):
    """grad_offset (v1) of a deformable conv with the column gradient kept in registers
    (kgdet_deform_conv_grad_offset)."""
    L = _lib.lib()
    _require_f32(input, offset, weight, grad_output)
    if packed is None:
        packed = pack_weight(weight.contiguous(), shape)
    grad_output = grad_output.contiguous()
    offset = offset.contiguous()
    grad_offset = torch.empty_like(offset)
    ws = _workspace(grad_output.device, L.kgdet_dcn_workspace_bytes(ctypes.byref(shape)))
    _lib.check(L.kgdet_deform_conv_grad_offset(
        ctypes.byref(shape), _lib.ptr(input.contiguous()), _lib.ptr(offset), _lib.ptr(packed), _lib.ptr(grad_output),
        _lib.ptr(grad_offset), ctypes.c_uint32(_lib.DCN_BF16 if bf16 else 0), _lib.ptr(ws),
        ctypes.c_size_t(ws.numel()), _lib.current_stream()), 'kgdet_deform_conv_grad_offset')
    return grad_offset


def grad_weights_grouped(inputs, offsets, grad_outputs, weights, shapes):
    """grad_weight of several v1 convs in one launch (kgdet_deform_conv_grad_weight_grouped); problem j uses
    inputs[j], offsets[j], grad_outputs[j] (the window described by shapes[j]) and returns a tensor like weights[j].
    Returns None when a problem is not eligible for the plane kernel."""
    L = _lib.lib()
    n = len(shapes)
    gws = [torch.empty_like(w, memory_format=torch.contiguous_format) for w in weights]
    arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    shape_arr = (ctypes.POINTER(_lib.DcnShape) * n)(*[ctypes.pointer(s) for s in shapes])
    ws = _workspace(inputs[0].device, L.kgdet_dcn_group_workspace_bytes(ctypes.c_int32(n), shape_arr))
    rc = L.kgdet_deform_conv_grad_weight_grouped(
        ctypes.c_int32(n), shape_arr, arr(inputs), arr(offsets), arr(grad_outputs), arr(gws), _lib.ptr(ws),
        ctypes.c_size_t(ws.numel()), _lib.current_stream())
    if rc == _lib.KGDET_E_UNSUPPORTED:
        return None
    _lib.check(rc, 'kgdet_deform_conv_grad_weight_grouped')
    return gws


def _backward(input, offset, mask, weight, bias, grad_output, shape, packed, needs):
    """Returns grad_input, grad_offset, grad_mask, grad_weight, grad_bias (None where not needed)."""
    L = _lib.lib()
    grad_output = grad_output.contiguous()
    ws_bytes = L.kgdet_dcn_workspace_bytes(ctypes.byref(shape))
    ws = _workspace(input.device, ws_bytes)
    grad_input = grad_offset = grad_mask = grad_weight = grad_bias = None
    if needs['input'] or needs['offset'] or needs['mask']:
        grad_input = torch.zeros_like(input)
        grad_offset = torch.empty_like(offset)
        grad_mask = torch.empty_like(mask) if mask is not None else None
        rc = L.kgdet_deform_conv_backward_input(
            ctypes.byref(shape), _lib.ptr(input), _lib.ptr(offset), _lib.ptr(mask), _lib.ptr(packed),
            _lib.ptr(grad_output), _lib.ptr(grad_input), _lib.ptr(grad_offset), _lib.ptr(grad_mask),
            _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.current_stream())
        if rc == _lib.KGDET_E_UNSUPPORTED and shape.deformable_groups == 1:
            # Maps beyond the LDS plane run on the column-gradient kernels, which want 16-row blocks of output channels per weight
            # group and an even number of (tap, channel) columns; the reference takes any channel counts
            # (deform_conv_cuda.cpp:260-371).  Such a call runs on the SAME deterministic kernels with zero channels appended per
            # weight group -- zero grad_output rows / zero input planes / zero weights contribute nothing to any gradient.
            grad_input, grad_offset, grad_mask = _backward_input_padded(input, offset, mask, weight, grad_output, shape)
        else:
            _lib.check(rc, 'kgdet_deform_conv_backward_input')
    if needs['weight'] or needs['bias']:
        grad_weight = torch.empty_like(weight, memory_format=torch.contiguous_format)
        grad_bias = torch.empty_like(bias) if bias is not None else None
        _lib.check(L.kgdet_deform_conv_backward_weight(
            ctypes.byref(shape), _lib.ptr(input), _lib.ptr(offset), _lib.ptr(mask),
            _lib.ptr(grad_output), _lib.ptr(grad_weight), _lib.ptr(grad_bias), 0,
            _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.current_stream()),
            'kgdet_deform_conv_backward_weight')
    return grad_input, grad_offset, grad_mask, grad_weight, grad_bias


def _backward_input_padded(input, offset, mask, weight, grad_output, shape):
    """grad_input / grad_offset / grad_mask of a convolution whose channel counts the large-map kernels do not take as they are:
    per weight group the output channels are padded to a multiple of 16 and the input channels to an even count (zeros)."""
    L = _lib.lib()
    G, kh, kw = shape.groups, weight.shape[2], weight.shape[3]
    N, C, H, W = input.shape
    O = weight.shape[0]
    Cg, Og = C // G, O // G
    Cp, Op = Cg + (Cg * kh * kw) % 2, -(-Og // 16) * 16

    def pad_groups(t, per, per_p, dim):      # [.., G * per, ..] -> [.., G * per_p, ..] with zeros behind every group
        if per == per_p:
            return t.contiguous()
        sh = list(t.shape)
        v = t.reshape(sh[:dim] + [G, per] + sh[dim + 1:])
        z = v.new_zeros(sh[:dim] + [G, per_p - per] + sh[dim + 1:])
        return torch.cat([v, z], dim + 1).reshape(sh[:dim] + [G * per_p] + sh[dim + 1:]).contiguous()
    xp, gp = pad_groups(input, Cg, Cp, 1), pad_groups(grad_output, Og, Op, 1)
    wp = weight.new_zeros(G, Op, Cp, kh, kw)
    wp[:, :Og, :Cg] = weight.reshape(G, Og, Cg, kh, kw)
    wp = wp.reshape(G * Op, Cp, kh, kw)
    shp = _shape(xp, wp, (shape.stride_h, shape.stride_w), (shape.pad_h, shape.pad_w), (shape.dil_h, shape.dil_w), G, 1)
    packed = pack_weight(wp, shp)
    ws = _workspace(input.device, L.kgdet_dcn_workspace_bytes(ctypes.byref(shp)))
    gi = torch.zeros_like(xp)
    go = torch.empty_like(offset)
    gm = torch.empty_like(mask) if mask is not None else None
    _lib.check(L.kgdet_deform_conv_backward_input(
        ctypes.byref(shp), _lib.ptr(xp), _lib.ptr(offset), _lib.ptr(mask), _lib.ptr(packed), _lib.ptr(gp), _lib.ptr(gi),
        _lib.ptr(go), _lib.ptr(gm), _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.current_stream()),
        'kgdet_deform_conv_backward_input (channels padded)')
    if Cp != Cg:
        gi = gi.reshape(N, G, Cp, H, W)[:, :, :Cg].reshape(N, C, H, W).contiguous()
    return gi, go, gm


class DeformConvFunction(Function):

    @staticmethod
    def forward(ctx, input, offset, weight, stride=1, padding=0, dilation=1, groups=1,
                deformable_groups=1, im2col_step=64, relu=False):
        if input is not None and input.dim() != 4:
            raise ValueError('Expected 4D tensor as input, got {}D tensor instead.'.format(input.dim()))
        ctx.stride = _pair(stride)
        ctx.padding = _pair(padding)
        ctx.dilation = _pair(dilation)
        ctx.groups = groups
        ctx.deformable_groups = deformable_groups
        ctx.im2col_step = im2col_step
        ctx.relu = relu
        if not input.is_cuda:
            raise NotImplementedError
        # the reference batches images in groups of im2col_step and insists it divides the batch
        # (R:47-49); the fused kernel needs no such batching but keeps the contract
        cur_im2col_step = min(ctx.im2col_step, input.shape[0])
        assert (input.shape[0] % cur_im2col_step) == 0, 'im2col step must divide batchsize'
        shape = _shape(input, weight, ctx.stride, ctx.padding, ctx.dilation, groups, deformable_groups)
        output, packed = _forward(input, offset, None, weight, None, shape, relu=relu)
        ctx.shape = shape
        ctx.save_for_backward(input, offset, weight, packed, output if relu else None)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, offset, weight, packed, output = ctx.saved_tensors
        if not grad_output.is_cuda:
            raise NotImplementedError
        if ctx.relu:
            grad_output = torch.ops.aten.threshold_backward(grad_output, output, 0)     # ReLU backward, one launch
        needs = dict(input=ctx.needs_input_grad[0], offset=ctx.needs_input_grad[1], mask=False,
                     weight=ctx.needs_input_grad[2], bias=False)
        gi, go, _, gw, _ = _backward(input.contiguous(), offset.contiguous(), None, weight, None,
                                     grad_output, ctx.shape, packed, needs)
        return (gi, go, gw, None, None, None, None, None, None, None)


class ModulatedDeformConvFunction(Function):

    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1,
                groups=1, deformable_groups=1):
        ctx.stride = stride
        ctx.padding = padding
        ctx.dilation = dilation
        ctx.groups = groups
        ctx.deformable_groups = deformable_groups
        ctx.with_bias = bias is not None
        if not input.is_cuda:
            raise NotImplementedError
        shape = _shape(input, weight, _pair(stride), _pair(padding), _pair(dilation), groups,
                       deformable_groups)
        output, packed = _forward(input, offset, mask, weight, bias, shape)
        ctx.shape = shape
        ctx.save_for_backward(input, offset, mask, weight, bias, packed)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        if not grad_output.is_cuda:
            raise NotImplementedError
        input, offset, mask, weight, bias, packed = ctx.saved_tensors
        needs = dict(input=True, offset=True, mask=True, weight=True, bias=ctx.with_bias)
        gi, go, gm, gw, gb = _backward(input.contiguous(), offset.contiguous(), mask.contiguous(),
                                       weight, bias, grad_output, ctx.shape, packed, needs)
        return (gi, go, gm, gw, gb, None, None, None, None, None)


def deform_conv(*args):
    """deform_conv(input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1,
    im2col_step=64) -- R:186"""
    return _autocast_apply(DeformConvFunction.apply, args)


def modulated_deform_conv(*args):
    """modulated_deform_conv(input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
    deformable_groups=1) -- R:187"""
    return _autocast_apply(ModulatedDeformConvFunction.apply, args)


def _fan_in_uniform_(weight, in_channels, kernel_size):
    """the reference modules' initialisation (deform_conv.py:227-232, :291-298): U(-1/sqrt(fan), +1/sqrt(fan)), fan = in_channels * kh * kw
    (the FULL in_channels, also for grouped weights -- kept as the reference has it)"""
    bound = 1.0 / math.sqrt(in_channels * kernel_size[0] * kernel_size[1])
    with torch.no_grad():
        weight.uniform_(-bound, bound)


class _DeformConvBase(nn.Module):
    """what the four op modules share: the constructor contract of the reference (argument names, attribute names, the
    ``weight`` / ``bias`` parameters and their checkpoint keys) and the weight-image cache rule"""

    def _configure(self, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, deformable_groups):
        for name, c in (('in_channels', in_channels), ('out_channels', out_channels)):
            if c % groups:
                raise AssertionError('%s = %d is not a multiple of groups = %d' % (name, c, groups))
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _pair(kernel_size)
        self.groups, self.deformable_groups = groups, deformable_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *self.kernel_size))
        return stride, padding, dilation

    def train(self, mode=True):
        clear_pack_cache()   # weights may have been updated without a version bump (fused optimizers)
        return super().train(mode)

    def reset_parameters(self):
        _fan_in_uniform_(self.weight, self.in_channels, self.kernel_size)
        if getattr(self, 'bias', None) is not None:
            nn.init.zeros_(self.bias)

    def _offset_conv(self, maps_per_tap):
        """the Pack variants' zero-initialised convolution that predicts the offsets (and masks) from the input itself"""
        conv = nn.Conv2d(self.in_channels, self.deformable_groups * maps_per_tap * self.kernel_size[0] * self.kernel_size[1],
                         kernel_size=self.kernel_size, stride=_pair(self.stride), padding=_pair(self.padding), bias=True)
        nn.init.zeros_(conv.weight)
        nn.init.zeros_(conv.bias)
        return conv


class DeformConv(_DeformConvBase):
    """mmdet/ops/dcn/deform_conv.py:190-236 (v1: no bias, stride / padding / dilation as pairs)"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, deformable_groups=1, bias=False):
        super(DeformConv, self).__init__()
        assert not bias, 'DeformConv has no bias (deform_conv.py:204)'
        stride, padding, dilation = self._configure(in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                                                    deformable_groups)
        self.stride, self.padding, self.dilation = _pair(stride), _pair(padding), _pair(dilation)
        self.reset_parameters()

    def forward(self, x, offset):
        return deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation,
                           self.groups, self.deformable_groups)


class DeformConvPack(DeformConv):
    """deform_conv.py:239-262: the offsets come from ``conv_offset(x)``"""

    def __init__(self, *args, **kwargs):
        super(DeformConvPack, self).__init__(*args, **kwargs)
        self.conv_offset = self._offset_conv(2)

    def init_offset(self):
        nn.init.zeros_(self.conv_offset.weight)
        nn.init.zeros_(self.conv_offset.bias)

    def forward(self, x):
        return deform_conv(x, self.conv_offset(x), self.weight, self.stride, self.padding, self.dilation,
                           self.groups, self.deformable_groups)


class ModulatedDeformConv(_DeformConvBase):
    """deform_conv.py:265-305 (v2: optional bias, stride / padding / dilation kept as given)"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, deformable_groups=1, bias=True):
        super(ModulatedDeformConv, self).__init__()
        self.stride, self.padding, self.dilation = self._configure(in_channels, out_channels, kernel_size, stride, padding,
                                                                   dilation, groups, deformable_groups)
        self.with_bias = bias
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def forward(self, x, offset, mask):
        return modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride,
                                     self.padding, self.dilation, self.groups,
                                     self.deformable_groups)


class ModulatedDeformConvPack(ModulatedDeformConv):
    """deform_conv.py:308-337: offsets and masks from ``conv_offset_mask(x)`` (first two thirds: offsets, last third: sigmoid masks)"""

    def __init__(self, *args, **kwargs):
        super(ModulatedDeformConvPack, self).__init__(*args, **kwargs)
        self.conv_offset_mask = self._offset_conv(3)

    def init_offset(self):
        nn.init.zeros_(self.conv_offset_mask.weight)
        nn.init.zeros_(self.conv_offset_mask.bias)

    def forward(self, x):
        both = self.conv_offset_mask(x)
        n_off = both.shape[1] // 3 * 2
        offset, mask = both[:, :n_off], torch.sigmoid(both[:, n_off:])
        return modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride,
                                     self.padding, self.dilation, self.groups,
                                     self.deformable_groups)


# ------------------------------------------------------------------------------------------------
# Fused multi-kernel deformable convolution:  relu(cat([dconv_k(x, offset_k, W_k) for k], dim=1))
# ------------------------------------------------------------------------------------------------
# bench.py's per-product timing sets the library's KGDET_OPT_BWD_PHASE switch; the grouped call then returns KGDET_E_PARTIAL (half of
# its outputs are not written).  Only a caller that declared a measurement accepts that; a training step raises.
MEASUREMENT = False
_SUM_IN_FIXUP = os.environ.get('KGDET_DCN_SUM_IN_FIXUP', '1') == '1'     # A/B switch: aliased gradient outputs, summed by the fix-up kernels


class DeformConvCatFunction(Function):
    """KGDet runs a 3x3, a 5x5 and a 7x7 deformable conv on a feature map, applies ReLU to each and
    concatenates them -- once on the classification features and once on the keypoint features, with
    the same offsets (reppoints_head_kp3rep_cas_1_assign_once.py:145-163).  Here
      * every conv writes its channel window of ONE output buffer per feature map with ReLU fused into
        the kernel epilogue (C ABI fields out_channel_offset / out_channels_total): no cat, no ReLU
        kernels, no slice copies, and backward reads the matching gradient windows in place;
      * all n_x * n_k convs of the call go to the GPU as ONE grouped launch
        (kgdet_deform_conv_forward_grouped): split-K slabs, fix-up and launch overhead are paid once.
    Stride 1, dilation 1, groups 1 (the head's configuration).
    args = x_0 .. x_{n_x-1}, offset_0 .. offset_{n_k-1}, then weights x-major: w[x][k].
    """

    @staticmethod
    def forward(ctx, relu, n_x, n_k, pads, *args):
        xs = [a.contiguous() for a in args[:n_x]]
        if not xs[0].is_cuda:
            raise NotImplementedError
        offsets = [a.contiguous() for a in args[n_x:n_x + n_k]]
        weights = args[n_x + n_k:]
        assert len(weights) == n_x * n_k
        _require_f32(*xs, *offsets, *weights)
        L = _lib.lib()
        N, C, H, W = xs[0].shape
        outs, shapes, wcs = [], [], []
        for i, x in enumerate(xs):
            ws_i = weights[i * n_k:(i + 1) * n_k]
            O_total = sum(w.shape[0] for w in ws_i)
            out = x.new_empty(N, O_total, H, W)
            o_base = 0
            for k in range(n_k):
                s = _shape(x, ws_i[k], (1, 1), (pads[k], pads[k]), (1, 1), 1, 1)
                s.out_channel_offset, s.out_channels_total = o_base, O_total
                _check_offset(offsets[k], s, (N, ws_i[k].shape[0], H, W))
                shapes.append(s)
                wcs.append(ws_i[k].contiguous())
                o_base += ws_i[k].shape[0]
            outs.append(out)
        split_only = split_images_suffice(shapes)
        packs = pack_weights(wcs, shapes, split_only)      # one launch for all of them
        n = n_x * n_k
        arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
        shape_arr = (ctypes.POINTER(_lib.DcnShape) * n)(*[ctypes.pointer(s) for s in shapes])
        ws = _workspace(xs[0].device, L.kgdet_dcn_group_workspace_bytes(ctypes.c_int32(n), shape_arr))
        _lib.check(L.kgdet_deform_conv_forward_grouped(
            ctypes.c_int32(n), shape_arr, arr([xs[j // n_k] for j in range(n)]),
            arr([offsets[j % n_k] for j in range(n)]), None, arr(packs), None,
            arr([outs[j // n_k] for j in range(n)]), _fwd_flags(relu), _lib.ptr(ws), ctypes.c_size_t(ws.numel()),
            _lib.current_stream()), 'kgdet_deform_conv_forward_grouped')
        ctx.shapes, ctx.relu, ctx.n_x, ctx.n_k, ctx.split_only = shapes, relu, n_x, n_k, split_only
        ctx.save_for_backward(*xs, *outs, *offsets, *weights, *packs)
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *grad_outs):
        n_x, n_k = ctx.n_x, ctx.n_k
        saved = ctx.saved_tensors
        xs, outs = saved[:n_x], saved[n_x:2 * n_x]
        offsets = saved[2 * n_x:2 * n_x + n_k]
        weights = saved[2 * n_x + n_k:2 * n_x + n_k + n_x * n_k]
        packs = saved[2 * n_x + n_k + n_x * n_k:]
        need = ctx.needs_input_grad[4:]
        grad_xs = [None] * n_x
        grad_offs = [None] * n_k
        grad_ws = [None] * (n_x * n_k)
        gouts = []
        for i in range(n_x):
            grad_out = grad_outs[i]
            if ctx.relu:
                grad_out = torch.ops.aten.threshold_backward(grad_out.contiguous(), outs[i], 0)   # ReLU backward, one launch
            gouts.append(grad_out.contiguous())
        need_io = any(need[:n_x + n_k])
        done_io = False
        if need_io and not _EXACT_BACKWARD:
            # grad_input / grad_offset of all n_x * n_k convs: two grouped launches
            L = _lib.lib()
            n = n_x * n_k
            arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
            shape_arr = (ctypes.POINTER(_lib.DcnShape) * n)(*[ctypes.pointer(s) for s in ctx.shapes])
            ws = _workspace(xs[0].device, L.kgdet_dcn_group_workspace_bytes(ctypes.c_int32(n), shape_arr))

            def grouped(gis, gos):
                return L.kgdet_deform_conv_backward_input_grouped(
                    ctypes.c_int32(n), shape_arr, arr([xs[j // n_k] for j in range(n)]),
                    arr([offsets[j % n_k] for j in range(n)]), arr(packs), arr([gouts[j // n_k] for j in range(n)]),
                    arr(gis), arr(gos), _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.current_stream())
            # ONE grad_input tensor per feature map and ONE grad_offset tensor per offset tensor: problems that share an output
            # pointer are summed into it by the kernels' fix-up (include/kgdet_hip.h) -- no stack + sum passes here
            gi_x, go_k = [torch.empty_like(x) for x in xs], [torch.empty_like(o) for o in offsets]
            rc = grouped([gi_x[j // n_k] for j in range(n)], [go_k[j % n_k] for j in range(n)]) if _SUM_IN_FIXUP else \
                _lib.KGDET_E_UNSUPPORTED
            if rc == _lib.KGDET_E_PARTIAL and MEASUREMENT:
                rc = _lib.KGDET_OK
            if rc == _lib.KGDET_OK:
                done_io = True
                for i in range(n_x):
                    if need[i]:
                        grad_xs[i] = gi_x[i]
                for k in range(n_k):
                    if need[n_x + k]:
                        grad_offs[k] = go_k[k]
            else:     # (shapes that do not tile alike / no static schedule: one output per problem, summed here)
                gis = [torch.empty_like(xs[j // n_k]) for j in range(n)]
                gos = [torch.empty_like(offsets[j % n_k]) for j in range(n)]
                rc = grouped(gis, gos)
                if rc == _lib.KGDET_OK:
                    done_io = True
                    for i in range(n_x):
                        if need[i]:
                            grad_xs[i] = torch.stack(gis[i * n_k:(i + 1) * n_k]).sum(0) if n_k > 1 else gis[i * n_k]
                    for k in range(n_k):
                        if need[n_x + k]:
                            grad_offs[k] = torch.stack(gos[k::n_k]).sum(0) if n_x > 1 else gos[k]
            if rc not in (_lib.KGDET_OK, _lib.KGDET_E_UNSUPPORTED):
                _lib.check(rc, 'kgdet_deform_conv_backward_input_grouped')
        done_w = False
        if all(need[n_x + n_k:]) and not _EXACT_BACKWARD:
            n = n_x * n_k
            gws = grad_weights_grouped([xs[j // n_k] for j in range(n)], [offsets[j % n_k] for j in range(n)],
                                       [gouts[j // n_k] for j in range(n)], weights, ctx.shapes)
            if gws is not None:
                grad_ws, done_w = gws, True
        for i in range(n_x):
            for k in range(n_k):
                j = i * n_k + k
                needs = dict(input=need[i] and not done_io, offset=need[n_x + k] and not done_io, mask=False,
                             weight=need[n_x + n_k + j] and not done_w, bias=False)
                if not any(needs.values()):
                    continue
                if ctx.split_only:     # (a fallback kernel after all: it reads the fp32 images the forward did not pack)
                    packs = pack_weights([w.contiguous() for w in weights], ctx.shapes)
                    ctx.split_only = False
                gi, go, _, gw, _ = _backward(xs[i], offsets[k], None, weights[j], None, gouts[i], ctx.shapes[j],
                                             packs[j], needs)
                if gi is not None and needs['input']:
                    grad_xs[i] = gi if grad_xs[i] is None else grad_xs[i].add_(gi)
                if go is not None and needs['offset']:
                    grad_offs[k] = go if grad_offs[k] is None else grad_offs[k].add_(go)
                if needs['weight']:
                    grad_ws[j] = gw
        return (None, None, None, None) + tuple(grad_xs) + tuple(grad_offs) + tuple(grad_ws)


def deform_conv_cat_multi(xs, offsets, weights, paddings, relu=True):
    """[relu(cat([deform_conv(x, o_k, w[i][k], 1, p_k) for k], dim=1)) for i, x in enumerate(xs)]
    as one grouped launch; weights[i][k] belongs to feature map i and kernel size k."""
    n_x, n_k = len(xs), len(offsets)
    flat_w = [w for ws in weights for w in ws]
    outs = _autocast_apply(DeformConvCatFunction.apply,
                           (relu, n_x, n_k, tuple(int(p) for p in paddings), *xs, *offsets, *flat_w))
    return list(outs)


def deform_conv_cat(x, offsets, weights, paddings, relu=True):
    """relu(cat([deform_conv(x, o, w, 1, p) for o, w, p in ...], dim=1)) in one buffer."""
    return deform_conv_cat_multi([x], offsets, [weights], paddings, relu)[0]
