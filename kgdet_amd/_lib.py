"""ctypes binding of libkgdet_hip.so (include/kgdet_hip.h).

The HIP library IS the product: there is no CPU or pure-PyTorch fallback behind it.  If the
shared object is missing or an entry point fails, the caller gets an exception.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('KGDET_LIB') or os.path.join(_HERE, 'libkgdet_hip.so')   # KGDET_LIB: experiment builds
CSRC = os.path.join(_HERE, 'csrc')

KGDET_OK = 0
KGDET_E_SHAPE = 1
KGDET_E_WORKSPACE = 2
KGDET_E_HIP = 3
KGDET_E_UNSUPPORTED = 4
KGDET_E_PARTIAL = 5

DCN_RELU = 1
DCN_BF16 = 2         # forward operands rounded to bf16 once (autocast inference)
DCN_EXACT_FP32 = 4   # forward on the exact-fp32 MFMA kernel instead of the bf16 hi/lo split


class DcnShape(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        'N', 'C', 'H', 'W', 'O', 'kh', 'kw', 'stride_h', 'stride_w', 'pad_h', 'pad_w', 'dil_h',
        'dil_w', 'groups', 'deformable_groups', 'out_channel_offset', 'out_channels_total')]


class PsroiShape(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        'B', 'C', 'H', 'W', 'R', 'out_dim', 'group_size', 'pooled_size', 'part_size',
        'sample_per_part', 'no_trans', 'num_classes')] + [
        ('spatial_scale', ctypes.c_float), ('trans_std', ctypes.c_float)]


def build(force=False):
    """Compile every HIP source for gfx950 into kgdet_amd/libkgdet_hip.so (hipcc, in-tree)."""
    cmd = ['make', '-C', CSRC, '-j8']
    if force:
        cmd.append('-B')
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def lib():
    """The loaded library.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                'kgdet_amd: %s is missing -- run `python -c "import __graft_entry__ as g; g.build()"` '
                'or `make -C kgdet_amd/csrc`. There is no non-HIP fallback.' % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.kgdet_last_error.restype = ctypes.c_char_p
        for name in ('kgdet_dcn_packed_weight_bytes', 'kgdet_dcn_workspace_bytes', 'kgdet_dcn_group_workspace_bytes',
                     'kgdet_nms_workspace_bytes', 'kgdet_deform_psroi_backward_workspace_bytes',
                     'kgdet_deform_psroi_forward_workspace_bytes', 'kgdet_head_loss_workspace_bytes',
                     'kgdet_moment_bbox_backward_workspace_bytes', 'kgdet_multiclass_soft_nms_workspace_bytes'):
            if hasattr(L, name):
                getattr(L, name).restype = ctypes.c_size_t
        if hasattr(L, 'kgdet_multiclass_soft_nms_supported'):
            L.kgdet_multiclass_soft_nms_supported.restype = ctypes.c_int
            L.kgdet_multiclass_soft_nms_supported.argtypes = [ctypes.c_int32] * 4
        _lib = L
    return _lib


def check(rc, what=''):
    """Map a status code to the exception the reference raises for the same condition."""
    if rc == KGDET_OK:
        return
    msg = lib().kgdet_last_error().decode('utf-8', 'replace')
    if rc == KGDET_E_UNSUPPORTED:
        raise NotImplementedError('%s: %s' % (what, msg))
    raise RuntimeError('%s: %s' % (what, msg))


def ptr(t):
    """device pointer of a tensor (or NULL for None)"""
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def raw_stream(device_index=None):
    """the current HIP stream of the device as an integer handle.  torch.cuda.current_stream() costs ~9 us of Python
    per call (a training step makes ~400 of them); the raw getter is a single C call."""
    import torch
    if device_index is None:
        device_index = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(device_index)


def current_stream():
    return ctypes.c_void_p(raw_stream())
