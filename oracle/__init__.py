"""CPU oracle for the KGDet hot path -- TEST INFRASTRUCTURE ONLY.

This package is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it; ``kgdet_amd`` never does.

It restates the reference's algorithms (R = /root/reference/mmdetection/mmdet) on numpy arrays:
the per-element "column" kernels live in C (``kgdet_oracle.c`` + ``*.inc``, compiled by
``make -C oracle``), and this module plays the role of the reference's C++ host functions that
put GEMMs between them (R/ops/dcn/src/deform_conv_cuda.cpp).  ``numpy.matmul`` stands in for
``at::addmm_``.

Pinning (see kgdet_oracle.c header): nms / soft_nms are pinned against golden vectors made by the
compiled reference; deform-conv / psroi / focal are "parity unpinned" by the reference itself
and are cross-checked against independent formulations in tests/test_oracle_*.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'libkgdet_oracle.so')


def build(force=False):
    """Compile the C part with gcc (idempotent)."""
    srcs = [os.path.join(_HERE, f) for f in ('kgdet_oracle.c', 'dcn_oracle.inc', 'psroi_oracle.inc')]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(['make', '-C', _HERE, '-s', '-B'])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_nms.restype = ctypes.c_int64
        _lib.oracle_soft_nms.restype = ctypes.c_int64
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _real(dtype):
    if dtype == np.float32:
        return 'f32', ctypes.c_float
    if dtype == np.float64:
        return 'f64', ctypes.c_double
    raise TypeError('oracle supports float32/float64, got %s' % dtype)


def conv_output_size(H, W, kh, kw, stride, padding, dilation):
    """R/ops/dcn/deform_conv.py:96-110."""
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dh, dw = _pair(dilation)
    return ((H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1,
            (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1)


# ---------------------------------------------------------------------------------------------
# column stages (thin ctypes shims)
# ---------------------------------------------------------------------------------------------
def dcn_columns(im, offset, mask, kh, kw, stride, padding, dilation, deformable_groups):
    """col[C*K, N, Ho, Wo] from im[N,C,H,W]; R/ops/dcn/src/deform_conv_cuda_kernel.cu:190-242 / 570-632."""
    sfx, _ = _real(im.dtype)
    N, C, H, W = im.shape
    sh, sw = _pair(stride); ph, pw = _pair(padding); dh, dw = _pair(dilation)
    Ho, Wo = conv_output_size(H, W, kh, kw, stride, padding, dilation)
    col = np.empty((C * kh * kw, N, Ho, Wo), im.dtype)
    getattr(lib(), 'oracle_dcn_columns_' + sfx)(
        _p(im), _p(offset), _p(mask), C, H, W, kh, kw, ph, pw, sh, sw, dh, dw, N,
        deformable_groups, _p(col))
    return col


def dcn_scatter_input(col, offset, mask, im_shape, kh, kw, stride, padding, dilation, deformable_groups):
    """grad_im[N,C,H,W] from col grads; deform_conv_cuda_kernel.cu:279-334 / 635-692."""
    sfx, _ = _real(col.dtype)
    N, C, H, W = im_shape
    sh, sw = _pair(stride); ph, pw = _pair(padding); dh, dw = _pair(dilation)
    grad_im = np.zeros(im_shape, col.dtype)
    getattr(lib(), 'oracle_dcn_scatter_input_' + sfx)(
        _p(col), _p(offset), _p(mask), C, H, W, kh, kw, ph, pw, sh, sw, dh, dw, N,
        deformable_groups, _p(grad_im))
    return grad_im


def dcn_offset_grad(col, im, offset, mask, kh, kw, stride, padding, dilation, deformable_groups):
    """(grad_offset, grad_mask|None); deform_conv_cuda_kernel.cu:373-435 / 695-766."""
    sfx, _ = _real(col.dtype)
    N, C, H, W = im.shape
    sh, sw = _pair(stride); ph, pw = _pair(padding); dh, dw = _pair(dilation)
    grad_offset = np.zeros_like(offset)
    grad_mask = np.zeros_like(mask) if mask is not None else None
    getattr(lib(), 'oracle_dcn_offset_grad_' + sfx)(
        _p(col), _p(im), _p(offset), _p(mask), C, H, W, kh, kw, ph, pw, sh, sw, dh, dw, N,
        deformable_groups, _p(grad_offset), _p(grad_mask))
    return grad_offset, grad_mask


# ---------------------------------------------------------------------------------------------
# deformable convolution v1 / v2: the reference's host algorithm (materialised columns + GEMM)
# ---------------------------------------------------------------------------------------------
def _c(a):
    return None if a is None else np.ascontiguousarray(a)


def deform_conv_forward(x, offset, weight, stride=1, padding=0, dilation=1, groups=1,
                        deformable_groups=1, mask=None, bias=None):
    """out[N,O,Ho,Wo].  v1: R/ops/dcn/src/deform_conv_cuda.cpp:151-258; v2 (mask[, bias]): :486-564.

    Columns are [C*K, N*Ho*Wo] (row = c*K + t); per group g the output block is
    ``weight[g].flatten(1) @ columns[g]`` (:229-234).  The reference's im2col_step batching only
    reorders the same products, so all images go through one GEMM here.
    """
    x, offset, weight, mask = _c(x), _c(offset), _c(weight), _c(mask)
    N, C, H, W = x.shape
    O, Cg, kh, kw = weight.shape
    assert C == Cg * groups and O % groups == 0
    Ho, Wo = conv_output_size(H, W, kh, kw, stride, padding, dilation)
    assert offset.shape == (N, deformable_groups * 2 * kh * kw, Ho, Wo), offset.shape
    col = dcn_columns(x, offset, mask, kh, kw, stride, padding, dilation, deformable_groups)
    col = col.reshape(groups, Cg * kh * kw, N * Ho * Wo)
    wg = weight.reshape(groups, O // groups, Cg * kh * kw)
    out = np.matmul(wg, col)                                   # [G, O/G, N*Ho*Wo]
    out = out.reshape(O, N, Ho, Wo).transpose(1, 0, 2, 3)
    if bias is not None:
        out = out + bias.reshape(1, -1, 1, 1)                   # :561-563
    return np.ascontiguousarray(out)


def deform_conv_backward(x, offset, weight, grad_out, stride=1, padding=0, dilation=1, groups=1,
                         deformable_groups=1, mask=None, with_bias=False):
    """Returns dict(grad_input, grad_offset, grad_weight[, grad_mask][, grad_bias]).

    v1: deform_conv_cuda.cpp:260-371 (input/offset) and :373-484 (weight, scale = 1);
    v2: :566-679.  colgrad = weight[g]^T @ grad_out[g] (:329-332), then offset_grad and
    scatter_input; grad_weight = grad_out[g] @ columns[g]^T (:456-462).
    """
    x, offset, weight, mask, grad_out = _c(x), _c(offset), _c(weight), _c(mask), _c(grad_out)
    N, C, H, W = x.shape
    O, Cg, kh, kw = weight.shape
    Ho, Wo = conv_output_size(H, W, kh, kw, stride, padding, dilation)
    K = kh * kw
    go = grad_out.transpose(1, 0, 2, 3).reshape(groups, O // groups, N * Ho * Wo)
    wg = weight.reshape(groups, O // groups, Cg * K)
    colgrad = np.matmul(wg.transpose(0, 2, 1), go)             # [G, Cg*K, N*Ho*Wo]
    colgrad = np.ascontiguousarray(colgrad.reshape(C * K, N, Ho, Wo))
    grad_offset, grad_mask = dcn_offset_grad(colgrad, x, offset, mask, kh, kw, stride, padding,
                                             dilation, deformable_groups)
    grad_input = dcn_scatter_input(colgrad, offset, mask, x.shape, kh, kw, stride, padding,
                                   dilation, deformable_groups)
    col = dcn_columns(x, offset, mask, kh, kw, stride, padding, dilation, deformable_groups)
    col = col.reshape(groups, Cg * K, N * Ho * Wo)
    grad_weight = np.matmul(go, col.transpose(0, 2, 1)).reshape(weight.shape)
    res = dict(grad_input=grad_input, grad_offset=grad_offset,
               grad_weight=np.ascontiguousarray(grad_weight))
    if mask is not None:
        res['grad_mask'] = grad_mask
    if with_bias:
        res['grad_bias'] = grad_out.sum(axis=(0, 2, 3))          # :659-665
    return res


# ---------------------------------------------------------------------------------------------
# deformable PS-RoI pooling
# ---------------------------------------------------------------------------------------------
def deform_psroi_forward(data, rois, offset, spatial_scale, out_size, out_channels, no_trans,
                         group_size=1, part_size=None, sample_per_part=4, trans_std=0.0):
    """(out, count); R/ops/dcn/src/deform_pool_cuda_kernel.cu:53-140, host :266-309."""
    data, rois = _c(data), _c(rois)
    sfx, creal = _real(data.dtype)
    part_size = out_size if part_size is None else part_size
    B, C, H, W = data.shape
    R = rois.shape[0]
    num_classes = 1 if no_trans else offset.shape[1] // 2
    offset = None if no_trans else _c(offset)
    out = np.empty((R, out_channels, out_size, out_size), data.dtype)
    cnt = np.empty_like(out)
    getattr(lib(), 'oracle_deform_psroi_forward_' + sfx)(
        _p(data), _p(rois), _p(offset), R, C, H, W, creal(spatial_scale), out_channels,
        group_size, out_size, part_size, sample_per_part, creal(trans_std), int(bool(no_trans)),
        num_classes, _p(out), _p(cnt))
    return out, cnt


def deform_psroi_backward(grad_out, count, data, rois, offset, spatial_scale, out_size,
                          out_channels, no_trans, group_size=1, part_size=None, sample_per_part=4,
                          trans_std=0.0):
    """(grad_data, grad_offset); deform_pool_cuda_kernel.cu:143-263."""
    grad_out, count, data, rois = _c(grad_out), _c(count), _c(data), _c(rois)
    sfx, creal = _real(data.dtype)
    part_size = out_size if part_size is None else part_size
    B, C, H, W = data.shape
    R = rois.shape[0]
    num_classes = 1 if no_trans else offset.shape[1] // 2
    grad_data = np.zeros_like(data)
    if no_trans:
        offset, grad_offset = None, None
    else:
        offset = _c(offset)
        grad_offset = np.zeros_like(offset)
    getattr(lib(), 'oracle_deform_psroi_backward_' + sfx)(
        _p(grad_out), _p(count), _p(data), _p(rois), _p(offset), R, C, H, W,
        creal(spatial_scale), out_channels, group_size, out_size, part_size, sample_per_part,
        creal(trans_std), int(bool(no_trans)), num_classes, _p(grad_data), _p(grad_offset))
    return grad_data, grad_offset


# ---------------------------------------------------------------------------------------------
# NMS / soft-NMS
# ---------------------------------------------------------------------------------------------
def nms(dets, iou_thr):
    """keep indices (int64, ascending); R/ops/nms/src/nms_cpu.cpp:5-59 (IoU >= thr suppresses)."""
    dets = np.ascontiguousarray(dets, np.float32)
    n = dets.shape[0]
    keep = np.empty(n, np.int64)
    m = lib().oracle_nms(_p(dets), ctypes.c_int64(n), ctypes.c_float(iou_thr), _p(keep))
    return keep[:m].copy()


def soft_nms(dets, iou_thr, method='linear', sigma=0.5, min_score=1e-3):
    """(new_dets[M,5] float32, inds[M] int64); R/ops/nms/src/soft_nms_cpu.pyx:22-127,
    wrapper R/ops/nms/nms_wrapper.py:52-78."""
    codes = {'linear': 1, 'gaussian': 2}
    if method not in codes:
        raise ValueError('Invalid method for SoftNMS: {}'.format(method))
    boxes = np.array(dets, np.float32, copy=True, order='C')
    n = boxes.shape[0]
    inds = np.empty(n, np.int64)
    m = lib().oracle_soft_nms(_p(boxes), ctypes.c_int64(n), ctypes.c_float(iou_thr),
                              codes[method], ctypes.c_float(sigma), ctypes.c_float(min_score),
                              _p(inds))
    return boxes[:m].copy(), inds[:m].copy()


# ---------------------------------------------------------------------------------------------
# sigmoid focal loss
# ---------------------------------------------------------------------------------------------
def sigmoid_focal_loss_forward(logits, targets, gamma=2.0, alpha=0.25):
    """loss[N,C]; R/ops/sigmoid_focal_loss/src/sigmoid_focal_loss_cuda.cu:24-59."""
    logits = np.ascontiguousarray(logits, np.float32)
    targets = np.ascontiguousarray(targets, np.int64)
    n, c = logits.shape
    out = np.empty_like(logits)
    lib().oracle_sigmoid_focal_loss_forward(_p(logits), _p(targets), ctypes.c_int64(n), c,
                                            ctypes.c_float(gamma), ctypes.c_float(alpha), _p(out))
    return out


def sigmoid_focal_loss_backward(logits, targets, d_losses, gamma=2.0, alpha=0.25):
    """d_logits[N,C]; sigmoid_focal_loss_cuda.cu:62-97."""
    logits = np.ascontiguousarray(logits, np.float32)
    targets = np.ascontiguousarray(targets, np.int64)
    d_losses = np.ascontiguousarray(d_losses, np.float32)
    n, c = logits.shape
    out = np.empty_like(logits)
    lib().oracle_sigmoid_focal_loss_backward(_p(logits), _p(targets), _p(d_losses),
                                             ctypes.c_int64(n), c, ctypes.c_float(gamma),
                                             ctypes.c_float(alpha), _p(out))
    return out
