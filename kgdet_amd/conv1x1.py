"""1x1 and 3x3 (stride 1, padding 1) convolutions, fp32 NCHW, on the bf16 hi/lo-split MFMA GEMM kernels
(csrc/conv1x1.hip).

``conv_split(x, weight)`` equals ``F.conv2d(x, weight, padding=k // 2)`` for a ``[O, C, k, k]`` weight, k in {1, 3},
to fp32-level accuracy (~5e-6 of the output scale), forward and both gradients; used by the backbone's bottlenecks
(kgdet_amd/backbone.py), where MIOpen's fp32 kernels run at 60-110 TFLOP/s."""
import ctypes
import weakref

import torch

from . import _lib


# The 3x3 grad_weight kernel (conv_nt8<9>): 95-104 us for the layer-2 / layer-3 shapes against MIOpen's implicit-GEMM
# wrw at 110 us + its layout transposes and output zeroing (~175 us in the step profile).  KGDET_SPLIT_WGRAD3=0 and
# KGDET_PACK_BOTH=0 switch back for A/B measurements.
import os as _os
SPLIT_GRAD_WEIGHT_3X3 = _os.environ.get('KGDET_SPLIT_WGRAD3', '1') == '1'
PACK_BOTH = _os.environ.get('KGDET_PACK_BOTH', '1') == '1'


PAD_GRAD_WEIGHT_3X3 = _os.environ.get('KGDET_NO_WPAD') is None   # (A/B switch for the zero-padded odd-width route)
# Maps with an odd pixel count (13 x 21, 7 x 11: the two coarsest levels of a five-level head) go to the split kernels as they
# are (round 4: tools/check_odd_maps.py -- forward 3e-7, gradients 5e-6 of torch's float64 convolution; 16-byte loads need 4-byte
# alignment only on gfx950).  0: rounds 2-3's route, one zero column appended and cut off around every convolution (pad_odd /
# unpad_odd below: six extra launches per convolution and step, ~130 of a config-5 step)
DIRECT_ODD_MAPS = _os.environ.get('KGDET_DIRECT_ODD_MAPS', '1') == '1'
ENABLED = True      # False: every dense convolution stays on MIOpen's fp32 kernels (dcn.arithmetic('exact'))


# 1x1 convolutions of ANY channel counts (round 6: the head's 13- / 588- / 166-channel output convolutions): the operand images pad
# a reduction that ends inside a 16-channel chunk with zeros, the kernel re-reads the last channel for them.  0: those stay with
# the vendor libraries (A/B).
RAGGED_1X1 = _os.environ.get('KGDET_CONV_RAGGED_1X1', '1') == '1'


def applicable(x, weight, stride=(1, 1), padding=(0, 0), dilation=(1, 1), groups=1):
    k = weight.shape[2]
    aligned = weight.shape[1] % 16 == 0 and weight.shape[0] % 16 == 0
    return (ENABLED and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4
            and weight.shape[2] == weight.shape[3] and k in (1, 3) and tuple(stride) == (1, 1)
            and tuple(padding) == (k // 2, k // 2) and tuple(dilation) == (1, 1) and groups == 1
            and x.is_contiguous() and (aligned or (k == 1 and RAGGED_1X1 and (weight.shape[0] * weight.shape[1]) % 2 == 0))
            and ((x.shape[2] * x.shape[3]) % 2 == 0 or DIRECT_ODD_MAPS) and x.shape[2] * x.shape[3] >= 4
            and not torch.is_autocast_enabled())


_L = None
_sizes = {}     # (query, args) -> bytes: the workspace / image sizes depend on the shape only


def _lib_sizes():
    """the library with argtypes declared once (plain Python ints / pointers marshal ~3x faster than c_int objects)"""
    global _L
    if _L is None:
        L = _lib.lib()
        vp, i32, i64, sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t
        for name, res, args in (
                ('kgdet_conv_packed_bytes', sz, [i32, i32, i32]),
                ('kgdet_conv_apply_workspace_bytes', sz, [i64, i32, i32, i32, i32, i32, i32]),
                ('kgdet_conv1x1_grad_weight_workspace_bytes', sz, [i64, i32, i32, i64]),
                ('kgdet_conv3x3_grad_weight_workspace_bytes', sz, [i64, i32, i32, i32, i32]),
                ('kgdet_conv_pack_fmt', ctypes.c_int, [vp, i32, i32, i32, i32, vp, i32, vp]),
                ('kgdet_conv_pack_both_fmt', ctypes.c_int, [vp, i32, i32, i32, vp, vp, i32, vp]),
                ('kgdet_conv_pack_blocks', i64, [i32, i32, i32]),
                ('kgdet_conv_pack_multi', ctypes.c_int, [vp, i32, i64, vp]),
                ('kgdet_conv_apply', ctypes.c_int, [vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, vp, sz, vp]),
                ('kgdet_conv_apply_epilogue_fmt', ctypes.c_int,
                 [vp, vp, vp, vp, vp, i32, i64, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp]),
                ('kgdet_conv_apply_gated_fmt', ctypes.c_int,
                 [vp, vp, vp, vp, vp, i32, vp, i64, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp]),
                ('kgdet_conv3x3_s2_grad_input', ctypes.c_int, [vp, vp, vp, i64, i32, i32, i32, i32, vp]),
                ('kgdet_conv3x3_s2_grad_weight_workspace_bytes', sz, [i64, i32, i32, i32, i32]),
                ('kgdet_conv3x3_s2_grad_weight', ctypes.c_int, [vp, vp, vp, i64, i32, i32, i32, i32, vp, sz, vp]),
                ('kgdet_conv1x1_grad_weight', ctypes.c_int, [vp, vp, vp, i64, i32, i32, i64, vp, sz, vp]),
                ('kgdet_conv1x1_grad_weight_fold', ctypes.c_int,
                 [vp, vp, vp, i64, i32, i32, i64, vp, sz, vp, vp, vp, vp, ctypes.c_float, vp, i32, vp, vp, vp]),
                ('kgdet_conv3x3_grad_weight_fold', ctypes.c_int,
                 [vp, vp, vp, i64, i32, i32, i32, i32, vp, sz, vp, vp, vp, vp, ctypes.c_float, vp, i32, vp, vp, vp]),
                ('kgdet_conv3x3_grad_weight', ctypes.c_int, [vp, vp, vp, i64, i32, i32, i32, i32, vp, sz, vp])):
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _L = L
    return _L


def _size(name, *args):
    key = (name, args)
    n = _sizes.get(key)
    if n is None:
        n = _sizes[key] = getattr(_lib_sizes(), name)(*args)
    return n


def _stream():
    return _lib.raw_stream()


# Operand format of the FORWARD images (activations x weights): two fp16 parts (22 mantissa bits, fp32-class results) instead of
# two bf16 parts (16 bits) -- csrc/conv1x1.hip split_pair_t.  The packed tensor carries the format as an attribute
# (`kgdet_f16`), `_apply` reads it; transposed (grad_input) images and the weight-gradient kernels stay bf16: gradients need
# the exponent range.  KGDET_CONV_FWD_F16=0: bf16 parts everywhere (A/B, and round 2's arithmetic).
FORWARD_F16 = _os.environ.get('KGDET_CONV_FWD_F16', '1') == '1'


def _mark(img, f16):
    img.kgdet_f16 = bool(f16)
    return img


def _pack(weight, transpose):
    """weight [O, C, k, k] -> operand image (forward: rows O; transpose: rows C with mirrored taps)"""
    L = _lib_sizes()
    O, C, taps = weight.shape[0], weight.shape[1], weight.shape[2] * weight.shape[3]
    M, K = (C, O) if transpose else (O, C)
    img = torch.empty(_size('kgdet_conv_packed_bytes', M, K, taps), dtype=torch.uint8, device=weight.device)
    f16 = FORWARD_F16 and not transpose
    _lib.check(L.kgdet_conv_pack_fmt(weight.data_ptr(), O, C, taps, 1 if transpose else 0, img.data_ptr(), 1 if f16 else 0,
                                     _stream()), 'conv_pack')
    return _mark(img, f16)


def _pack_both(weight):
    """forward and grad_input images of one weight in one launch"""
    L = _lib_sizes()
    O, C, taps = weight.shape[0], weight.shape[1], weight.shape[2] * weight.shape[3]
    img = torch.empty(_size('kgdet_conv_packed_bytes', O, C, taps), dtype=torch.uint8, device=weight.device)
    img_t = torch.empty(_size('kgdet_conv_packed_bytes', C, O, taps), dtype=torch.uint8, device=weight.device)
    _lib.check(L.kgdet_conv_pack_both_fmt(weight.data_ptr(), O, C, taps, img.data_ptr(), img_t.data_ptr(),
                                          1 if FORWARD_F16 else 0, _stream()), 'conv_pack_both')
    return _mark(img, FORWARD_F16), _mark(img_t, False)


def _apply(img, x, M, taps, stride=1, bias=None, residual=None, relu=False, gate=None):
    """y = conv(x) through the packed image; inference epilogue [relu](y + bias [+ residual]) fused into the store;
    ``gate`` (shape of y): y is zeroed where gate <= 0 (kgdet_conv_apply_gated_fmt: a ReLU's backward in the store)"""
    L = _lib_sizes()
    B, K, H, W = x.shape
    y = torch.empty((B, M, (H + stride - 1) // stride, (W + stride - 1) // stride), dtype=torch.float32, device=x.device)
    nbytes = _size('kgdet_conv_apply_workspace_bytes', B, M, K, H, W, taps, stride)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
    if gate is not None and (gate.shape != y.shape or gate.dtype != torch.float32 or not gate.is_contiguous()):
        raise ValueError('gate must be a contiguous fp32 tensor of the output\'s shape')
    _lib.check(L.kgdet_conv_apply_gated_fmt(
        img.data_ptr(), x.data_ptr(), y.data_ptr(), bias.data_ptr() if bias is not None else None,
        residual.data_ptr() if residual is not None else None, 1 if relu else 0,
        gate.data_ptr() if gate is not None else None, B, M, K, H, W, taps, stride,
        1 if getattr(img, 'kgdet_f16', False) else 0, ws.data_ptr() if nbytes else None, nbytes, _stream()), 'conv_apply')
    return y


def gate_applicable(y_shape):
    """can `_apply(..., gate=)` serve an output of this shape?  (the K-split sum handles pixel pairs)"""
    return (y_shape[2] * y_shape[3]) % 2 == 0


# ---- one pack launch per training step --------------------------------------------------------------------------------
# Training re-packs every weight every step (the optimizer changed it): 60 launches of a few microseconds.  Inside a
# `step_scope()` (the detector's forward_train) the weights seen in earlier steps are packed TOGETHER, into persistent
# image buffers, by one kgdet_conv_pack_multi launch at scope entry; a weight met for the first time is packed on its own
# and joins the set.  Outside a scope nothing is cached: every call packs (weights may change between any two calls).
class _Entry(object):
    __slots__ = ('ref', 'img', 'img_t', 'token', 'ptr')


_entries = {}        # id(weight) -> _Entry
_token = 0           # current scope generation; 0 = no scope active
_generation = 0
_table = None        # (key, device descriptor tensor, total blocks)
PACK_MULTI = _os.environ.get('KGDET_PACK_MULTI', '1') == '1'


# ---- frozen-statistics BatchNorm folded into the convolution in front of it (kgdet_amd/backbone.py _ConvBNActFold) ------
# A (weight, BatchNorm) pair seen inside a step scope joins `_fold_entries`; from the next scope on the scope's pack launch
# writes the images of w * s, s = gamma / sqrt(var + eps), and the pair's s and t = beta - mean * s are refreshed before it by a
# handful of multi-tensor torch ops over flat buffers (the parameters change every step).
FOLD_BN = _os.environ.get('KGDET_FOLD_BN', '1') == '1'


class _FoldEntry(object):
    __slots__ = ('ref', 'bn', 'img', 'img_t', 's', 't', 'token', 'ptr')


_fold_entries = {}   # id(weight) -> _FoldEntry
_fold_flat = None    # (key, S, T, TMP, EPS, s views, t views, tmp views, gammas, betas, means, vars)


def _fold_refresh(live):
    """s and t of every folded pair, into flat buffers whose slices the entries hold"""
    global _fold_flat
    # (the BatchNorm's parameter / buffer OBJECTS are part of the key: replacing bn.weight by a new Parameter, or a buffer by
    #  load_state_dict(assign=True), must not leave the old tensors' values in the folded images)
    key = tuple((k, e.ptr, id(e.bn()), id(e.bn().weight), id(e.bn().bias), id(e.bn().running_mean), id(e.bn().running_var),
                 e.bn().weight.data_ptr(), e.bn().running_var.data_ptr()) for k, e in live)
    if _fold_flat is None or _fold_flat[0] != key:
        dev = live[0][1].img.device
        sizes = [e.ref().shape[0] for _, e in live]
        total = sum(sizes)
        S, T, TMP = (torch.empty(total, dtype=torch.float32, device=dev) for _ in range(3))
        EPS = torch.cat([torch.full((n,), float(e.bn().eps), dtype=torch.float32) for n, (_, e) in zip(sizes, live)]).to(dev)
        sv, tv, mv, off = [], [], [], 0
        for n, (_, e) in zip(sizes, live):
            sv.append(S[off:off + n]); tv.append(T[off:off + n]); mv.append(TMP[off:off + n])
            e.s, e.t = sv[-1], tv[-1]
            off += n
        bns = [e.bn() for _, e in live]
        _fold_flat = (key, S, T, TMP, EPS, sv, tv, mv, [b.weight for b in bns], [b.bias for b in bns],
                      [b.running_mean for b in bns], [b.running_var for b in bns])
    _, S, T, TMP, EPS, sv, tv, mv, gammas, betas, means, vars_ = _fold_flat
    with torch.no_grad():
        torch._foreach_copy_(mv, vars_)
        TMP.add_(EPS).rsqrt_()                        # 1 / sqrt(var + eps)
        torch._foreach_copy_(sv, gammas)
        S.mul_(TMP)                                   # s = gamma * invstd
        torch._foreach_copy_(tv, means)
        T.mul_(S).neg_()
        torch._foreach_add_(tv, betas)                # t = beta - mean * s


def _launch_multi():
    global _table
    dead = [k for k, e in _entries.items() if e.ref() is None or e.ref().data_ptr() != e.ptr]
    for k in dead:
        del _entries[k]
    dead = [k for k, e in _fold_entries.items() if e.ref() is None or e.bn() is None or e.ref().data_ptr() != e.ptr]
    for k in dead:
        del _fold_entries[k]
    for k in _fold_entries:            # a folded pair's plain images (its first step ran unfolded) are not needed any more
        _entries.pop(k, None)
    if not _entries and not _fold_entries:
        return
    live = list(_entries.items())
    folded = list(_fold_entries.items())
    if folded:
        _fold_refresh(folded)
    # the rows hold raw device pointers: the key names them too (ids alone are reused by CPython once a model is freed)
    key = (tuple((k, e.ptr, e.img.data_ptr(), e.img_t.data_ptr()) for k, e in live),
           tuple((k, e.ptr, e.img.data_ptr(), e.img_t.data_ptr(), e.s.data_ptr()) for k, e in folded))
    L = _lib_sizes()
    if _table is None or _table[0] != key:
        rows, first = [], 0
        for _, e in live + folded:
            w = e.ref()
            O, C, taps = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
            # (bit 62 of the fifth word: the forward image in fp16 parts; sixth word: per-output-channel scale or 0)
            rows.append([w.data_ptr(), e.img.data_ptr(), e.img_t.data_ptr(), (O << 32) | C,
                         (taps << 32) | first | ((1 << 62) if getattr(e.img, 'kgdet_f16', False) else 0),
                         e.s.data_ptr() if isinstance(e, _FoldEntry) else 0])
            first += L.kgdet_conv_pack_blocks(O, C, taps)
        dev = (live + folded)[0][1].img.device
        _table = (key, torch.tensor(rows, dtype=torch.int64).to(dev), first)   # (one upload per change of the set)
    _lib.check(L.kgdet_conv_pack_multi(_table[1].data_ptr(), len(live) + len(folded), _table[2], _stream()), 'conv_pack_multi')
    for _, e in live + folded:
        e.token = _token


def fold_images(weight, bn):
    """(forward image, grad_input image, s, t) of conv(., weight) followed by the frozen-statistics BatchNorm ``bn``, packed by
    the current step scope's launch -- or None: outside a scope, or the pair is new (it joins the set for the next scope)"""
    if not (FOLD_BN and PACK_MULTI and _token):
        return None
    e = _fold_entries.get(id(weight))
    if e is not None and e.ref() is weight and e.ptr == weight.data_ptr() and e.bn() is bn:
        return (e.img, e.img_t, e.s, e.t) if e.token == _token else None
    O, C, taps = weight.shape[0], weight.shape[1], weight.shape[2] * weight.shape[3]
    if (isinstance(weight, torch.nn.Parameter) and weight.is_contiguous() and O % 16 == 0 and C % 16 == 0 and bn.affine
            and taps in (1, 9)):
        e = _FoldEntry()
        e.ref, e.bn, e.ptr, e.token = weakref.ref(weight), weakref.ref(bn), weight.data_ptr(), 0
        e.img = _mark(torch.empty(_size('kgdet_conv_packed_bytes', O, C, taps), dtype=torch.uint8, device=weight.device),
                      FORWARD_F16)
        e.img_t = _mark(torch.empty(_size('kgdet_conv_packed_bytes', C, O, taps), dtype=torch.uint8, device=weight.device), False)
        e.s = e.t = None
        _fold_entries[id(weight)] = e
    return None


class step_scope(object):
    """``with conv1x1.step_scope():`` around ONE training forward (its backward may run after the scope closes: the
    images live in persistent buffers that are only rewritten by the next scope's pack launch)."""

    def __enter__(self):
        global _token, _generation
        self.prev = _token
        _generation += 1
        _token = _generation
        if PACK_MULTI and (_entries or _fold_entries):
            _launch_multi()

    def __exit__(self, *exc):
        global _token
        _token = self.prev


# proxy tensor -> the Parameter whose storage it shares (layers.shared_levels: the levels of a shared-weight head run on per-level
# leaf aliases of the parameters; the persistent operand images belong to the parameter)
_aliases = {}        # id(proxy) -> (weakref to the proxy, weakref to the parameter); entries leave with their proxy


def alias(proxy, param):
    key = id(proxy)
    _aliases[key] = (weakref.ref(proxy, lambda _r, key=key: _aliases.pop(key, None)), weakref.ref(param))


def forward_images(x, weight):
    """(forward operand image, grad_input operand image or None) of a contiguous weight"""
    both = PACK_BOTH and x.requires_grad and (weight.shape[0] % 16 == 0 or weight.shape[2] == 1)
    a = _aliases.get(id(weight))
    if a is not None and a[0]() is weight:
        origin = a[1]()
        if origin is not None and origin.data_ptr() == weight.data_ptr() and origin.shape == weight.shape:
            weight = origin      # (same storage: the images are the parameter's)
    if both and PACK_MULTI and _token:
        e = _entries.get(id(weight))
        if e is not None and e.ref() is weight and e.ptr == weight.data_ptr():
            if e.token != _token:       # joined the set after this scope's pack launch
                _lib.check(_lib_sizes().kgdet_conv_pack_both_fmt(weight.data_ptr(), weight.shape[0], weight.shape[1],
                                                                 weight.shape[2] * weight.shape[3], e.img.data_ptr(),
                                                                 e.img_t.data_ptr(), 1 if getattr(e.img, 'kgdet_f16', False) else 0,
                                                                 _stream()), 'conv_pack_both')
                e.token = _token
            return e.img, e.img_t
        img, img_t = _pack_both(weight)
        if isinstance(weight, torch.nn.Parameter):     # persistent tensors only
            e = _Entry()
            e.ref, e.img, e.img_t, e.token, e.ptr = weakref.ref(weight), img, img_t, _token, weight.data_ptr()
            _entries[id(weight)] = e
        return img, img_t
    if both:
        return _pack_both(weight)       # the backward's operand image comes out of the same launch
    return _pack(weight, False), None


def grad_weight(x, weight, gy):
    """grad of ``conv(x, weight)`` (stride 1, padding k // 2) with respect to the weight"""
    O, C, k = weight.shape[0], weight.shape[1], weight.shape[2]
    L = _lib_sizes()
    if k == 3 and SPLIT_GRAD_WEIGHT_3X3 and C % 128 == 0:
        if x.shape[3] % 4 and not PAD_GRAD_WEIGHT_3X3:
            return torch.nn.grad.conv2d_weight(x, weight.shape, gy, padding=1)
        # (W % 4 != 0, the 25 x 42 head / FPN maps: the library pads both operands inside its workspace, one launch --
        # the split kernel beats MIOpen's fp32 Winograd weight gradient, 42 against 76 us per call, and is deterministic)
        B, H, W = x.shape[0], x.shape[2], x.shape[3]
        nbytes = _size('kgdet_conv3x3_grad_weight_workspace_bytes', B, O, C, H, W)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        gw = torch.empty_like(weight)
        _lib.check(L.kgdet_conv3x3_grad_weight(gy.data_ptr(), x.data_ptr(), gw.data_ptr(), B, O, C, H, W,
                                               ws.data_ptr(), nbytes, _stream()), 'conv3x3_grad_weight')
        return gw
    if k != 1 or ((x.shape[2] * x.shape[3]) % 4 != 0 and not PAD_GRAD_WEIGHT_3X3):
        # other 3x3 shapes: MIOpen.  (1x1 on maps with H*W % 4 != 0, e.g. 25 x 42: the library pads both operands)
        return torch.nn.grad.conv2d_weight(x, weight.shape, gy, padding=k // 2)
    B, HW = x.shape[0], x.shape[2] * x.shape[3]
    nbytes = _size('kgdet_conv1x1_grad_weight_workspace_bytes', B, O, C, HW)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    gw = torch.empty_like(weight)
    _lib.check(L.kgdet_conv1x1_grad_weight(gy.data_ptr(), x.data_ptr(), gw.data_ptr(), B, O, C, HW,
                                           ws.data_ptr(), nbytes, _stream()), 'conv1x1_grad_weight')
    return gw


def grad_weight_fold_route(x, weight):
    """3 / 1: grad_weight_fold serves this problem with the 3x3 / 1x1 split kernels; 0: it returns None"""
    C, k = weight.shape[1], weight.shape[2]
    if k == 3 and SPLIT_GRAD_WEIGHT_3X3 and C % 128 == 0 and (x.shape[3] % 4 == 0 or PAD_GRAD_WEIGHT_3X3):
        return 3
    if k == 1 and ((x.shape[2] * x.shape[3]) % 4 == 0 or PAD_GRAD_WEIGHT_3X3):
        return 1
    return 0


def grad_weight_fold(x, weight, gy, s, mean, var, eps, bn_partial, P, want_gamma=True):
    """grad_weight of a convolution with a folded BatchNorm (backbone._ConvBNActFold): (s * G, sums [2, O] = grad_beta,
    grad_gamma) from ONE launch behind the split kernel -- or None where grad_weight would take another route.
    ``bn_partial=None`` (P = 0): the per-channel sums of gy are formed inside the weight-gradient kernel."""
    O, C, k = weight.shape[0], weight.shape[1], weight.shape[2]
    L = _lib_sizes()
    if bn_partial is None:
        P = 0

        class bn_partial(object):        # (a null pointer for the two calls below)
            @staticmethod
            def data_ptr():
                return None
    if k == 3 and SPLIT_GRAD_WEIGHT_3X3 and C % 128 == 0 and (x.shape[3] % 4 == 0 or PAD_GRAD_WEIGHT_3X3):
        B, H, W = x.shape[0], x.shape[2], x.shape[3]
        nbytes = _size('kgdet_conv3x3_grad_weight_workspace_bytes', B, O, C, H, W)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        gw = torch.empty_like(weight)
        sums = torch.empty((2, O), dtype=torch.float32, device=x.device)
        _lib.check(L.kgdet_conv3x3_grad_weight_fold(gy.data_ptr(), x.data_ptr(), gw.data_ptr(), B, O, C, H, W, ws.data_ptr(), nbytes,
                                                    weight.data_ptr(), s.data_ptr(), mean.data_ptr(), var.data_ptr(), eps,
                                                    bn_partial.data_ptr(), P, sums[0].data_ptr(),
                                                    sums[1].data_ptr() if want_gamma else None, _stream()), 'conv3x3_grad_weight_fold')
        return gw, sums
    if k == 1 and ((x.shape[2] * x.shape[3]) % 4 == 0 or PAD_GRAD_WEIGHT_3X3):
        B, HW = x.shape[0], x.shape[2] * x.shape[3]
        nbytes = _size('kgdet_conv1x1_grad_weight_workspace_bytes', B, O, C, HW)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        gw = torch.empty_like(weight)
        sums = torch.empty((2, O), dtype=torch.float32, device=x.device)
        _lib.check(L.kgdet_conv1x1_grad_weight_fold(gy.data_ptr(), x.data_ptr(), gw.data_ptr(), B, O, C, HW, ws.data_ptr(), nbytes,
                                                    weight.data_ptr(), s.data_ptr(), mean.data_ptr(), var.data_ptr(), eps,
                                                    bn_partial.data_ptr(), P, sums[0].data_ptr(),
                                                    sums[1].data_ptr() if want_gamma else None, _stream()), 'conv1x1_grad_weight_fold')
        return gw, sums
    return None


def grad_input(weight, img_t, gy, residual=None, gate=None):
    """grad of ``conv(x, weight)`` with respect to x [+ residual: another gradient of x, added in the kernel's store]
    [zeroed where gate <= 0: x = relu(.) and gate = x applies that ReLU's backward in the same store]"""
    C, k = weight.shape[1], weight.shape[2]
    return _apply(img_t if img_t is not None else _pack(weight, True), gy, C, k * k, residual=residual, gate=gate)


class _ConvSplit(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        weight = weight.contiguous()
        img, ctx.img_t = forward_images(x, weight)
        ctx.save_for_backward(x, weight)
        return _apply(img, x, weight.shape[0], weight.shape[2] * weight.shape[3])

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = grad_input(weight, ctx.img_t, gy) if ctx.needs_input_grad[0] else None
        gw = grad_weight(x, weight, gy) if ctx.needs_input_grad[1] else None
        return gx, gw


def conv_split(x, weight):
    return _ConvSplit.apply(x, weight)


PAD_ODD_MAPS = _os.environ.get('KGDET_PAD_ODD_MAPS', '1') == '1'     # 0: maps with an odd pixel count stay on MIOpen (A/B)


def odd_map_applicable(x, weight, stride=(1, 1), padding=(0, 0), dilation=(1, 1), groups=1):
    """a map with an odd number of pixels (13 x 21, 7 x 11: the two coarsest levels of a five-level head) that the split kernels
    take once ONE zero column is appended: for a 1x1 / 3x3 stride-1 convolution with its own zero padding the appended column
    reads as that padding, so the first W output columns are the convolution of the unpadded map.  (MIOpen's fp32 Winograd
    kernel costs ~47 us per pass whatever the map size: 47 launches, 1.8 ms of a config-5 step.)"""
    return (PAD_ODD_MAPS and x.dim() == 4 and (x.shape[2] * x.shape[3]) % 2 == 1 and x.shape[2] % 2 == 1 and x.is_cuda
            and x.dtype == torch.float32 and x.is_contiguous()
            and applicable(_PadProbe(x), weight, stride, padding, dilation, groups))


class _PadProbe(object):
    """what `applicable` asks of a tensor, for x with one more column (no allocation)"""

    def __init__(self, x):
        self.is_cuda, self.dtype = x.is_cuda, x.dtype
        self.shape = (x.shape[0], x.shape[1], x.shape[2], x.shape[3] + 1)

    def dim(self):
        return 4

    def is_contiguous(self):
        return True


def pad_odd(x):
    return torch.nn.functional.pad(x, (0, 1))


def unpad_odd(y, width):
    return y[..., :width].contiguous()


conv1x1 = conv_split


BIAS_GRAD_IN_WGRAD = _os.environ.get('KGDET_BIAS_GRAD_IN_WGRAD', '1') == '1'     # 0: gy.sum((0, 2, 3)) as its own reduce launch (A/B)
_ones_cache = {}


def _ones(n, device):
    t = _ones_cache.get((n, device))
    if t is None:
        t = _ones_cache[(n, device)] = torch.ones(n, dtype=torch.float32, device=device)
    return t


class _ConvBiasAct(torch.autograd.Function):
    """``[relu](conv(x, weight) + bias)`` with bias and ReLU in the convolution's store (the plain biased 3x3 convolutions of
    the head's first stage, KP3:69-71: MIOpen's fp32 Winograd takes 62 us for each of their three passes at 25 x 42)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        weight = weight.contiguous()
        img, ctx.img_t = forward_images(x, weight)
        y = _apply(img, x, weight.shape[0], weight.shape[2] * weight.shape[3], 1, bias.contiguous(), None, relu)
        ctx.relu = relu
        ctx.save_for_backward(x, weight, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        if ctx.relu:
            gy = torch.ops.aten.threshold_backward(gy, y, 0)
        gy = gy.contiguous()
        gx = grad_input(weight, ctx.img_t, gy) if ctx.needs_input_grad[0] else None
        if BIAS_GRAD_IN_WGRAD and ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and grad_weight_fold_route(x, weight):
            # the bias gradient = the row sums of grad_y, which the weight-gradient kernel forms on its way (the folded-BatchNorm
            # variant with s = 1: grad_w = 1 * G, grad_beta = the row sums) -- instead of one more pass over grad_y per convolution
            # (45 reduce launches of a config-5 step, 11 of a KGDet step)
            one = _ones(weight.shape[0], weight.device)
            res = grad_weight_fold(x, weight, gy, one, one, one, 0.0, None, 0, want_gamma=False)
            if res is not None:
                return gx, res[0], res[1][0], None
        gw = grad_weight(x, weight, gy) if ctx.needs_input_grad[1] else None
        gb = gy.sum((0, 2, 3)) if ctx.needs_input_grad[2] else None
        return gx, gw, gb, None


CACHE_INFER_CASTS = _os.environ.get('KGDET_CACHE_INFER_CASTS', '1') == '1'     # 0: autocast casts weights and biases per batch (A/B)


GEMM_1X1_NCHW = _os.environ.get('KGDET_INFER_GEMM_1X1_NCHW', '1') == '1'     # 0: MIOpen for the head's 1x1 output convolutions (A/B)


# module -> (key, reduced-precision weight, bias).  Weak keys, module-level: the copies are neither deep-copied nor pickled with
# the module (they used to sit in conv.__dict__).
_cast_cache = weakref.WeakKeyDictionary()


def invalidate_inference_caches():
    """Drop every derived inference-time copy of the weights: the autocast cast cache here, the folded conv + BatchNorm weights
    (backbone._fold_cache, the stem pack) and the packed deformable operands (dcn._pack_cache).  The caches follow the
    parameters' version counters, data pointers and object identities; a write THROUGH ``.data`` (``p.data.copy_``, EMA /
    weight-averaging utilities, some checkpoint loaders) changes none of the three -- callers that do that must call this.
    ``checkpoint.load_checkpoint`` and the detector's ``train()`` / ``eval()`` (detector.py: every switch of mode, so that weights
    stepped by a replayed HIP graph -- ``runner.GraphedTrainStep`` -- are never evaluated through stale copies) do; ``ResNet.train()``
    alone clears the folded-backbone cache only."""
    global _fold_flat
    _cast_cache.clear()
    _fold_flat = None
    from . import backbone, dcn
    backbone.clear_fold_cache()
    dcn.clear_pack_cache()


def conv_infer(conv, x):
    """``conv(x)``; inference under autocast keeps the reduced-precision copies of weight and bias across batches (autocast's own
    cache ends with its context: a batch re-cast the head's nine 3x3 weights, the 1x1 output weights and every bias -- ~30 launches,
    ~120 us of a 7 ms batch).  The copies follow the parameters' version counters."""
    if (CACHE_INFER_CASTS and type(conv) is torch.nn.Conv2d and not torch.is_grad_enabled() and x.is_cuda
            and torch.is_autocast_enabled() and conv.weight.dtype == torch.float32 and conv.padding_mode == 'zeros'):
        dt = torch.get_autocast_dtype('cuda')
        # channels-last activations meet a channels-last weight: MIOpen then runs its NHWC kernel as it is, without the
        # layout-conversion launches it wraps around an NCHW call (36 per batch of the KGDet head)
        cl = x.dim() == 4 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)
        key = (conv.weight._version, conv.weight.data_ptr(), id(conv.weight),
               None if conv.bias is None else (conv.bias._version, conv.bias.data_ptr(), id(conv.bias)), dt, cl)
        c = _cast_cache.get(conv)
        if c is None or c[0] != key:
            w = conv.weight.detach().to(dt)
            if cl:
                w = w.contiguous(memory_format=torch.channels_last)
            c = (key, w, None if conv.bias is None else conv.bias.detach().to(dt))
            _cast_cache[conv] = c
        if (GEMM_1X1_NCHW and not cl and x.dim() == 4 and x.is_contiguous() and x.dtype == dt and conv.kernel_size == (1, 1)
                and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.groups == 1):
            # a 1x1 convolution of an NCHW tensor (the deformable stages' outputs) IS W [Cout, Cin] @ x[b] [Cin, H*W]: one
            # batched GEMM in place, where MIOpen converts the activation to NHWC, convolves and converts back (+ a bias pass)
            B, _, H, W = x.shape
            y = torch.matmul(c[1].view(c[1].shape[0], -1), x.view(B, x.shape[1], H * W))
            if c[2] is not None:
                y += c[2].view(1, -1, 1)
            return y.view(B, -1, H, W)
        return torch.nn.functional.conv2d(x, c[1], c[2], conv.stride, conv.padding, conv.dilation, conv.groups)
    return conv(x)


def conv_bias_act(conv, x, relu=False):
    """``[relu](conv(x))`` of a plain ``nn.Conv2d``: fp32 training on the GPU takes the split-bf16 MFMA kernels, anything else
    the module itself (+ F.relu)"""
    if (type(conv) is torch.nn.Conv2d and conv.bias is not None and torch.is_grad_enabled()
            and conv.bias.dtype == torch.float32
            and applicable(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups)):
        return _ConvBiasAct.apply(x, conv.weight, conv.bias, relu)
    if (type(conv) is torch.nn.Conv2d and conv.bias is not None and torch.is_grad_enabled() and conv.bias.dtype == torch.float32
            and odd_map_applicable(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups)):
        return unpad_odd(_ConvBiasAct.apply(pad_odd(x), conv.weight, conv.bias, relu), x.shape[3])
    y = conv_infer(conv, x)
    return torch.relu(y) if relu else y


class _ConvSplitStride2(torch.autograd.Function):
    """3x3 stride-2 padding-1 convolution: forward on conv_nn<9> (MIOpen's fp32 strided kernels run at 15-20 TFLOP/s),
    grad_input on conv3x3_s2_grad_input (four parity classes), grad_weight on a gather of the nine strided views + the 1x1
    weight-gradient GEMM (round 6); MIOpen behind the switches."""

    @staticmethod
    def forward(ctx, x, weight):
        weight = weight.contiguous()
        img, ctx.img_t = forward_images(x, weight)
        ctx.save_for_backward(x, weight)
        return _apply(img, x, weight.shape[0], 9, 2)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = None
        if ctx.needs_input_grad[0] and STRIDE2_GRAD_INPUT and weight.shape[0] % 16 == 0:
            # the four parity classes of the output pixels on the patch kernel (csrc/conv1x1.hip conv3x3_s2_grad_input)
            img_t = ctx.img_t if ctx.img_t is not None else _pack(weight, True)
            gx = torch.empty_like(x)
            _lib.check(_lib_sizes().kgdet_conv3x3_s2_grad_input(
                img_t.data_ptr(), gy.data_ptr(), gx.data_ptr(), x.shape[0], x.shape[1], weight.shape[0], x.shape[2], x.shape[3],
                _stream()), 'conv3x3_s2_grad_input')
        need_gx = ctx.needs_input_grad[0] and gx is None
        gw = None
        need_gw = ctx.needs_input_grad[1]
        if need_gw and STRIDE2_GRAD_WEIGHT and (weight.shape[0] * weight.shape[1]) % 2 == 0:
            # the nine strided views of x gathered once, then the 1x1 weight-gradient GEMM over (tap, channel) columns
            # (csrc/conv1x1.hip kgdet_conv3x3_s2_grad_weight): MIOpen's igemm_wrw + its layout transposes were the last vendor
            # kernels of the training step
            L = _lib_sizes()
            B, C, H, W = x.shape
            O = weight.shape[0]
            nbytes = _size('kgdet_conv3x3_s2_grad_weight_workspace_bytes', B, O, C, H, W)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            gw = torch.empty_like(weight)
            _lib.check(L.kgdet_conv3x3_s2_grad_weight(gy.data_ptr(), x.data_ptr(), gw.data_ptr(), B, O, C, H, W, ws.data_ptr(), nbytes,
                                                      _stream()), 'conv3x3_s2_grad_weight')
            need_gw = False
        if need_gx or need_gw:
            gx2, gw2, _ = torch.ops.aten.convolution_backward(
                gy, x, weight, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [need_gx, need_gw, False])
            gx = gx2 if need_gx else gx
            gw = gw2 if need_gw else gw
        return gx, gw


STRIDE2 = _os.environ.get('KGDET_CONV_S2', '1') == '1'
STRIDE2_GRAD_INPUT = _os.environ.get('KGDET_CONV_S2_GI', '1') == '1'    # 0: MIOpen for the stride-2 grad_input (A/B)
STRIDE2_GRAD_WEIGHT = _os.environ.get('KGDET_CONV_S2_GW', '1') == '1'   # 0: MIOpen for the stride-2 grad_weight (A/B)


def applicable_stride2(x, weight, stride, padding, dilation, groups):
    return (ENABLED and STRIDE2 and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4
            and tuple(weight.shape[2:]) == (3, 3) and tuple(stride) == (2, 2) and tuple(padding) == (1, 1)
            and tuple(dilation) == (1, 1) and groups == 1 and x.is_contiguous() and weight.shape[1] % 16 == 0
            # (an odd number of OUTPUT pixels -- 25 x 42 -> 13 x 21 -> 7 x 11, config 5's two extra FPN levels -- in place like the
            #  stride-1 kernels since round 6: MIOpen's split-K forward for them adds with float atomics, and everything computed on
            #  those two levels differed in the last bits from run to run)
            and (((x.shape[2] + 1) // 2) * ((x.shape[3] + 1) // 2) % 2 == 0 or DIRECT_ODD_MAPS)
            and ((x.shape[2] + 1) // 2) * ((x.shape[3] + 1) // 2) >= 4 and not torch.is_autocast_enabled())


def conv3x3_stride2(x, weight):
    return _ConvSplitStride2.apply(x, weight)
