"""Checkpoint I/O with the reference's key names (mmcv ``load_checkpoint`` / ``save_checkpoint``
slices the KGDet tools use: T/train.py:94-100, T/test.py:174-180).  A checkpoint is
``dict(meta=..., state_dict=..., optimizer=...)``; DataParallel's ``module.`` prefix is stripped."""
import os
import time
from collections import OrderedDict

import torch


def _strip_module(state_dict):
    return OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in state_dict.items())


def load_state_dict(module, state_dict, strict=False):
    own = module.state_dict()
    unexpected, mismatched = [], []
    for name, param in state_dict.items():
        if name not in own:
            unexpected.append(name)
            continue
        if own[name].shape != param.shape:
            mismatched.append('{}: checkpoint {} vs model {}'.format(name, tuple(param.shape), tuple(own[name].shape)))
            continue
        own[name].copy_(param)
    missing = sorted(set(own.keys()) - set(state_dict.keys()))
    msgs = []
    if unexpected:
        msgs.append('unexpected key in source state_dict: {}'.format(', '.join(unexpected)))
    if missing:
        msgs.append('missing keys in source state_dict: {}'.format(', '.join(missing)))
    if mismatched:
        msgs.append('size mismatch: {}'.format('; '.join(mismatched)))
    if msgs and strict:
        raise RuntimeError('\n'.join(msgs))
    return dict(unexpected=unexpected, missing=missing, mismatched=mismatched)


def load_checkpoint(model, filename, map_location=None, strict=False):
    if not os.path.isfile(filename):
        raise IOError('{} is not a checkpoint file'.format(filename))
    checkpoint = torch.load(filename, map_location=map_location, weights_only=False)
    if isinstance(checkpoint, OrderedDict):
        state_dict = checkpoint
    elif isinstance(checkpoint, dict) and 'state_dict' in checkpoint:
        state_dict = checkpoint['state_dict']
    else:
        raise RuntimeError('No state_dict found in checkpoint file {}'.format(filename))
    target = model.module if hasattr(model, 'module') else model
    with torch.no_grad():
        load_state_dict(target, _strip_module(state_dict), strict)
    # derived weight images (folded conv+BN weights, packed deformable-conv operands) belong to the old weights
    from . import conv1x1
    conv1x1.invalidate_inference_caches()
    return checkpoint


def save_checkpoint(model, filename, optimizer=None, meta=None):
    meta = dict(meta or {})
    meta.update(time=time.asctime())
    target = model.module if hasattr(model, 'module') else model
    checkpoint = {'meta': meta,
                  'state_dict': OrderedDict((k, v.cpu()) for k, v in target.state_dict().items())}
    if optimizer is not None:
        checkpoint['optimizer'] = optimizer.state_dict()
    os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
    torch.save(checkpoint, filename)
