"""Gradient clipping + Adam as two multi-tensor HIP passes (csrc/optim.hip) for ``DistOptimizerHook``.

The reference's step is mmcv's ``OptimizerHook.after_train_iter`` as re-implemented in ``mmdet/core/utils/dist_utils.py:44-58``:
``clip_grad_norm_(params, max_norm=35, norm_type=2)`` then ``optimizer.step()``; the KGDet config trains with ``torch.optim.Adam``.
The fused path keeps torch's objects -- the optimizer's own ``exp_avg`` / ``exp_avg_sq`` / ``step`` state (checkpoints are
unchanged), the parameters' ``.grad`` (scaled in place when the clip is active, as ``clip_grad_norm_`` does) -- and only replaces
the ~15 launches between them.  Anything it does not cover (other optimizers, amsgrad, maximize, sparse or non-fp32 tensors,
the very first step, which creates the state) goes through torch.
"""
import ctypes
import math
import os

import torch

from . import _lib

ENABLED = os.environ.get('KGDET_FUSED_CLIP_ADAM', '1') == '1'      # 0: clip_grad_norm_ + optimizer.step() (A/B)


class FusedClipAdam(object):
    """one instance per (hook, optimizer): caches the pointer table of the step's tensors"""

    def __init__(self):
        self._key = None          # tuple of the pointers of the last table
        self._table = None        # device int64 [n, 6]
        self._partial = None
        self._norm = None
        self._pinned = None
        self._event = None
        self._host_step = None
        self._step_ref = None
        self._n_steps = 0
        # device-side schedule (the step inside a captured HIP graph): see enable_device_schedule
        self._sched = None        # device float [4]: lr, 1 - beta1^t, sqrt(1 - beta2^t), t
        self._lr_ring = None      # page-locked float [LR_RING]: slot k % LR_RING = learning rate of step k (1-based)
        self._dev_steps = 0       # steps taken through the device schedule (host mirror of sched[3])
        self._pending = 0         # ... of which not yet added to the optimizer's own step tensors
        self._ring_events = []

    LR_RING = 64

    def enable_device_schedule(self, optimizer):
        """From now on the step count, the bias corrections and the learning rate of the fused step live in device memory
        (csrc/optim.hip adam_schedule_step): the step can be captured in a HIP graph and replayed.  The host's part per step is
        ``publish_lr(optimizer)`` BEFORE the launch (replay) of step k: it writes that step's learning rate into its ring slot.
        ``sync_optimizer_state(optimizer)`` adds the steps taken to the optimizer's own ``step`` tensors (state_dict / checkpoint
        compatibility; called by the graphed runner before anything reads them)."""
        group = optimizer.param_groups[0]
        # (optimizer.state is a defaultdict: indexing a parameter without state would CREATE an empty entry -- and raise KeyError
        # on 'step'; a parameter that has a gradient but no state yet has not been stepped: not this schedule's case)
        stateless = [p for p in group['params'] if p.grad is not None and p not in optimizer.state]
        if stateless:
            raise RuntimeError('%d parameters have a gradient but no optimizer state: take one step through the optimizer first'
                               % len(stateless))
        steps = [optimizer.state[p]['step'] for p in group['params'] if p in optimizer.state and 'step' in optimizer.state[p]]
        steps = [s for s in steps if isinstance(s, torch.Tensor)]
        both = torch.stack([s.detach().reshape(()).float().cpu() for s in steps]).aminmax()
        if float(both.min) != float(both.max):
            raise RuntimeError('per-parameter step counters differ: the fused step does not apply')
        t = int(round(float(both.min)))
        dev = group['params'][0].device
        beta1, beta2 = group['betas']
        self._sched = torch.tensor([float(group['lr']), 1.0 - beta1 ** max(t, 1), math.sqrt(1.0 - beta2 ** max(t, 1)), float(t)],
                                   dtype=torch.float32, device=dev)
        self._lr_ring = torch.full((self.LR_RING,), float(group['lr']), dtype=torch.float32).pin_memory()
        self._dev_steps, self._pending = t, 0
        self._ring_events = [None] * self.LR_RING
        self._host_step = None
        self._sched_rows = len(steps)          # the set the one device-side counter stands for
        return self

    def disable_device_schedule(self, optimizer):
        """back to host-side counters (the hook's torch fallback is about to step the optimizer itself): the steps taken under the
        device schedule are added to the optimizer's own counters first, so the two never diverge"""
        if self._sched is not None:
            self.sync_optimizer_state(optimizer)
            self._sched = None
        self._host_step = None

    def publish_lr(self, optimizer):
        """host side of step ``self._dev_steps + 1`` under the device schedule: its learning rate into its ring slot (waiting,
        if the host ran a whole ring ahead, for the step that last read the slot)"""
        k = self._dev_steps + 1
        slot = k % self.LR_RING
        ev = self._ring_events[slot]
        if ev is not None:
            ev.synchronize()
        self._lr_ring[slot] = float(optimizer.param_groups[0]['lr'])

    def step_published(self):
        """after the launch (replay) of the step whose rate ``publish_lr`` wrote"""
        self._dev_steps += 1
        self._pending += 1
        ev = torch.cuda.Event()
        ev.record()
        self._ring_events[self._dev_steps % self.LR_RING] = ev

    def sync_optimizer_state(self, optimizer):
        if self._pending:
            group = optimizer.param_groups[0]
            steps = [optimizer.state[p]['step'] for p in group['params'] if p in optimizer.state]
            torch._foreach_add_(steps, float(self._pending))
            self._pending = 0

    def invalidate(self):
        """A step was taken outside this object (torch's ``optimizer.step()``): the host copy of the step count is stale."""
        self._host_step = None

    @staticmethod
    def applicable(optimizer, params, grad_clip):
        if not ENABLED or type(optimizer) is not torch.optim.Adam:
            return False
        if grad_clip is not None and float(grad_clip.get('norm_type', 2)) != 2.0:
            return False
        for group in optimizer.param_groups:
            if group.get('amsgrad') or group.get('maximize') or group.get('differentiable') or group.get('capturable') \
                    or group.get('decoupled_weight_decay'):      # (AdamW's update is a different expression)
                return False
            if isinstance(group['lr'], torch.Tensor):
                return False
        for p in params:
            if p.grad is None:
                continue
            st = optimizer.state.get(p)
            if (not st or 'exp_avg' not in st or not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous()
                    or p.grad.dtype != torch.float32 or p.grad.is_sparse or not p.grad.is_contiguous()
                    or not isinstance(st['step'], torch.Tensor)):
                return False
        return True

    def step(self, optimizer, params, grad_clip):
        """clip (over ``params``, the hook's list) + Adam (the optimizer's single parameter group).  Returns True when the
        step was taken (``last_norm`` = the total norm as a device scalar, or None), False when it must go through
        torch (per-parameter step counters differ)."""
        L = _lib.lib()
        chunk = L.kgdet_optim_chunk()
        group = optimizer.param_groups[0]
        clip_set = set(id(p) for p in params if p.grad is not None)
        rows, steps, first = [], [], 0
        for p in group['params']:
            if p.grad is None:
                continue
            st = optimizer.state[p]
            n = p.numel()
            rows.append((p.data_ptr(), p.grad.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), n, first))
            first += (n + chunk - 1) // chunk
            steps.append(st['step'])
            if grad_clip is not None and id(p) not in clip_set:
                raise RuntimeError('a stepped parameter is missing from the clipped set')
        if not rows:
            return True
        key = tuple(rows)
        dev = group['params'][0].device       # (the step counters live on the CPU unless the optimizer is fused / capturable)
        capturing = torch.cuda.is_current_stream_capturing()
        if key != self._key:
            # (the gradients are fresh tensors every step, but the caching allocator hands out the same blocks for the same
            #  sequence of requests: the table is uploaded again only when an address moved)
            if self._event is not None and not capturing:
                self._event.synchronize()        # the previous upload has left the pinned buffer
            if self._pinned is None or self._pinned.shape[0] < len(rows):
                self._pinned = torch.empty((len(rows), 6), dtype=torch.int64).pin_memory()
            host = self._pinned[:len(rows)]
            host.copy_(torch.tensor(rows, dtype=torch.int64))
            self._table = host.to(dev, non_blocking=True)
            if capturing:      # (a copy node of the graph, re-run by every replay from the same page-locked rows: keep them)
                self._pinned = None
                self._capture_rows = host
                self._event = None
            else:
                self._event = torch.cuda.Event()
                self._event.record()
            self._key = key
            if self._partial is None or self._partial.numel() < first:
                self._partial = torch.empty(first, dtype=torch.float32, device=dev)
                self._norm = torch.empty(1, dtype=torch.float32, device=dev)
        stream = _lib.current_stream()
        max_norm = 0.0
        if grad_clip is not None:
            max_norm = float(grad_clip.get('max_norm', 35))
            _lib.check(L.kgdet_multi_grad_norm(ctypes.c_void_p(self._table.data_ptr()), ctypes.c_int32(len(rows)),
                                               ctypes.c_int64(first), _lib.ptr(self._partial), _lib.ptr(self._norm), stream),
                       'multi_grad_norm')
        if self._sched is not None:
            # device schedule: nothing of the step is read or advanced on the host (the call may be a graph capture); the
            # caller publishes the learning rate and counts the steps (publish_lr / step_published).  ONE counter stands for every
            # stepped tensor: a parameter that joins (or leaves) would get the wrong bias correction -- refuse loudly
            if len(rows) != getattr(self, '_sched_rows', len(rows)):
                raise RuntimeError('the set of stepped parameters changed under the device-side Adam schedule (%d -> %d tensors): '
                                   'call disable_device_schedule() / enable_device_schedule() around such a change'
                                   % (self._sched_rows, len(rows)))
            beta1, beta2 = group['betas']
            _lib.check(L.kgdet_multi_clip_adam_dev(
                ctypes.c_void_p(self._table.data_ptr()), ctypes.c_int32(len(rows)), ctypes.c_int64(first), _lib.ptr(self._norm),
                ctypes.c_float(max_norm), _lib.ptr(self._sched), ctypes.c_void_p(self._lr_ring.data_ptr()),
                ctypes.c_int32(self.LR_RING), ctypes.c_double(beta1), ctypes.c_double(beta2), ctypes.c_float(group['eps']),
                ctypes.c_float(group['weight_decay']), stream), 'multi_clip_adam_dev')
            if not capturing:
                torch.autograd.graph.increment_version([p for p in group['params'] if p.grad is not None])
            self.last_norm = self._norm[0] if grad_clip is not None else None
            return True
        if self._host_step is None or steps[0] is not self._step_ref or len(steps) != self._n_steps:
            # ONE read-back, at the first fused step -- and again whenever the optimizer's state was replaced
            # (load_state_dict on resume creates new step tensors), the set of stepped tensors changed, or a step went
            # through torch in between (`invalidate()`, called by the hook's fallback branch).  The kernel applies one
            # bias correction to all tensors, torch keeps a counter per parameter: counters that differ (a parameter
            # that joined later) are not this kernel's case -- the caller falls back to optimizer.step().
            both = torch.stack([s.detach().reshape(()).float() for s in steps]).aminmax()
            lo, hi = float(both.min), float(both.max)
            if lo != hi:
                self._host_step = None
                return False
            self._host_step = int(round(lo))
            self._step_ref = steps[0]
            self._n_steps = len(steps)
        torch._foreach_add_(steps, 1.0)        # the optimizer's own step counters (state_dict compatibility)
        self._host_step += 1
        t = self._host_step
        beta1, beta2 = group['betas']
        _lib.check(L.kgdet_multi_clip_adam(
            ctypes.c_void_p(self._table.data_ptr()), ctypes.c_int32(len(rows)), ctypes.c_int64(first), _lib.ptr(self._norm),
            ctypes.c_float(max_norm), ctypes.c_float(group['lr']), ctypes.c_double(beta1), ctypes.c_double(beta2),
            ctypes.c_float(group['eps']), ctypes.c_float(group['weight_decay']),
            ctypes.c_float(1.0 - beta1 ** t), ctypes.c_float(math.sqrt(1.0 - beta2 ** t)), stream), 'multi_clip_adam')
        # the kernel wrote the parameters through raw pointers: tell autograd's version counters, so weight images
        # cached by `_version` (backbone fold / stem caches, DeformConv packs, conv1x1 tables) are rebuilt
        torch.autograd.graph.increment_version([p for p in group['params'] if p.grad is not None])
        self.last_norm = self._norm[0] if grad_clip is not None else None
        return True
