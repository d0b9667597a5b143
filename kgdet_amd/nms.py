"""NMS / soft-NMS -- host-side mirror of the reference's ``mmdet/ops/nms/nms_wrapper.py``.

R = mmdetection/mmdet/ops/nms/nms_wrapper.py: ``nms`` :8-49, ``soft_nms`` :52-78.  Same call
signatures and return conventions (torch in -> torch out, numpy in -> numpy out; kept indices in
ascending index order).  Both always run the HIP kernels (kgdet_amd/csrc/nms.hip): CPU tensors
and numpy arrays are staged to the current GPU, there is no host implementation.
``nms_batched`` is the extension the detector uses: every (image, class) group of one batch in a
single launch with no host synchronisation inside.
"""
import ctypes

import numpy as np
import torch

from . import _lib

_METHODS = {'linear': 1, 'gaussian': 2}


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError('kgdet_amd.nms needs a GPU: the HIP kernel is the only implementation')
    return torch.device('cuda', torch.cuda.current_device())


def nms_batched(dets, seg_offsets, iou_thr, max_seg_len=None):
    """dets [T,5] float32 (GPU); seg_offsets [S+1] int64 (GPU), segment i = rows [o[i], o[i+1]).

    Returns (keep [T] int64 -- segment-relative kept indices packed at each segment's offset,
    num_keep [S] int64).  No host sync when ``max_seg_len`` is given.
    """
    L = _lib.lib()
    S = seg_offsets.numel() - 1
    T = dets.shape[0]
    keep = torch.empty(T, dtype=torch.int64, device=dets.device)
    num_keep = torch.zeros(max(S, 1), dtype=torch.int64, device=dets.device)
    if S <= 0 or T == 0:
        return keep, num_keep[:max(S, 0)]
    if max_seg_len is None:
        max_seg_len = int((seg_offsets[1:] - seg_offsets[:-1]).max().item())
    dets = dets.contiguous().float()
    ws, ws_bytes = None, 0
    if max_seg_len > 4096:      # beyond the on-chip limit the kernel keeps its arrays in a scratch buffer
        ws_bytes = L.kgdet_nms_workspace_bytes(ctypes.c_int64(T), ctypes.c_int32(S))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dets.device)
    _lib.check(L.kgdet_nms_batched(_lib.ptr(dets), _lib.ptr(seg_offsets), ctypes.c_int32(S),
                                   ctypes.c_int64(T), ctypes.c_int64(max_seg_len), ctypes.c_float(iou_thr),
                                   _lib.ptr(keep), _lib.ptr(num_keep), _lib.ptr(ws), ctypes.c_size_t(ws_bytes),
                                   _lib.current_stream()), 'kgdet_nms_batched')
    return keep, num_keep[:S]


def nms(dets, iou_thr, device_id=None):
    """R:8-49.  Returns (dets[inds], inds)."""
    if isinstance(dets, torch.Tensor):
        is_numpy = False
        dets_th = dets
    elif isinstance(dets, np.ndarray):
        is_numpy = True
        dets_th = torch.from_numpy(dets)
    else:
        raise TypeError('dets must be either a Tensor or numpy array, but got {}'.format(type(dets)))

    if dets_th.shape[0] == 0:
        inds = dets_th.new_zeros(0, dtype=torch.long)
    else:
        if dets_th.is_cuda:
            dev = dets_th.device
        elif device_id is not None:
            dev = torch.device('cuda', device_id)
        else:
            dev = _device()
        with torch.cuda.device(dev):
            d = dets_th.detach().to(dev, torch.float32).contiguous()
            offs = torch.tensor([0, d.shape[0]], dtype=torch.int64, device=dev)
            keep, num = nms_batched(d, offs, float(iou_thr), max_seg_len=d.shape[0])
            inds = keep[:int(num.item())]
        inds = inds.to(dets_th.device)

    if is_numpy:
        inds = inds.cpu().numpy()
    return dets[inds, :], inds


def soft_nms(dets, iou_thr, method='linear', sigma=0.5, min_score=1e-3):
    """R:52-78.  Returns (new_dets, inds)."""
    if isinstance(dets, torch.Tensor):
        is_tensor = True
        dets_th = dets.detach()
    elif isinstance(dets, np.ndarray):
        is_tensor = False
        dets_th = torch.from_numpy(dets)
    else:
        raise TypeError('dets must be either a Tensor or numpy array, but got {}'.format(type(dets)))
    if method not in _METHODS:
        raise ValueError('Invalid method for SoftNMS: {}'.format(method))

    dev = dets_th.device if dets_th.is_cuda else _device()
    n = dets_th.shape[0]
    with torch.cuda.device(dev):
        d = dets_th.to(dev, torch.float32).contiguous()
        out = torch.empty(n, 5, dtype=torch.float32, device=dev)
        inds = torch.empty(n, dtype=torch.int64, device=dev)
        num = torch.zeros(1, dtype=torch.int64, device=dev)
        _lib.check(_lib.lib().kgdet_soft_nms(
            _lib.ptr(d), ctypes.c_int64(n), ctypes.c_float(iou_thr), ctypes.c_int32(_METHODS[method]),
            ctypes.c_float(sigma), ctypes.c_float(min_score), _lib.ptr(out), _lib.ptr(inds), _lib.ptr(num),
            _lib.current_stream()), 'kgdet_soft_nms')
        m = int(num.item())
    new_dets, inds = out[:m], inds[:m]
    if is_tensor:
        return new_dets.to(dets.device, dets.dtype), inds.to(dets.device)
    return new_dets.cpu().numpy().astype(np.float32), inds.cpu().numpy().astype(np.int64)
