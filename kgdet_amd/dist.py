"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

Reference behaviour (mmdet/core/utils/dist_utils.py:9-58): after ``backward()`` returns, all
gradients are flattened into ONE fp32 tensor (209 MB for KGDet), all-reduced, divided by the
world size, copied back, then clipped and applied -- nothing overlaps.

``allreduce_grads`` keeps that function (same signature, same result).  ``OverlappedGradReducer``
is the MI355X design: parameters are bucketed in reverse registration order (roughly the order
autograd finishes them: head first, backbone last); a post-accumulate-grad hook fires a bucket's
all-reduce on a side HIP stream as soon as its last gradient is ready, so RCCL traffic over the
xGMI links hides under the backbone's backward.  The division by the world size is folded into the
bucket copy.  Parameters that never receive a gradient (the unused FPN2 branches) are
found on the first step and excluded, like the reference's ``param.grad is not None`` filter.
"""
from collections import OrderedDict

import torch
import torch.distributed as dist
from torch._utils import _flatten_dense_tensors, _take_tensors, _unflatten_dense_tensors


def _allreduce_coalesced(tensors, world_size, bucket_size_mb=-1):
    if bucket_size_mb > 0:
        buckets = _take_tensors(tensors, bucket_size_mb * 1024 * 1024)
    else:
        by_type = OrderedDict()
        for tensor in tensors:
            by_type.setdefault(tensor.type(), []).append(tensor)
        buckets = by_type.values()
    for bucket in buckets:
        flat = _flatten_dense_tensors(bucket)
        dist.all_reduce(flat)
        flat.div_(world_size)
        for tensor, synced in zip(bucket, _unflatten_dense_tensors(flat, bucket)):
            tensor.copy_(synced)


def allreduce_grads(params, coalesce=True, bucket_size_mb=-1):
    grads = [param.grad.data for param in params if param.requires_grad and param.grad is not None]
    world_size = dist.get_world_size()
    if coalesce:
        _allreduce_coalesced(grads, world_size, bucket_size_mb)
    else:
        for tensor in grads:
            dist.all_reduce(tensor.div_(world_size))


def clip_grads(params, max_norm=35, norm_type=2):
    params = [p for p in params if p.requires_grad and p.grad is not None]
    if params:
        return torch.nn.utils.clip_grad_norm_(params, max_norm=max_norm, norm_type=norm_type)


class OverlappedGradReducer(object):
    """Bucketed all-reduce launched from gradient hooks, overlapped with the rest of backward."""

    def __init__(self, params, bucket_size_mb=32, process_group=None):
        self.group = process_group
        self.world_size = dist.get_world_size(process_group)
        self.params = [p for p in params if p.requires_grad]
        self.bucket_bytes = int(bucket_size_mb * 1024 * 1024)
        self.active = None          # params known to receive gradients (learned on the first step)
        self.buckets = None
        self.use_cuda = len(self.params) > 0 and self.params[0].is_cuda
        self.stream = torch.cuda.Stream() if self.use_cuda else None
        self._hooks = []
        self._pending = []

    # -- bucket construction ----------------------------------------------------------------------
    def _build(self):
        order = list(reversed(self.active))
        self.buckets, cur, cur_bytes = [], [], 0
        for p in order:
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > self.bucket_bytes or cur[0].dtype != p.dtype):
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {}
        self._flat = []
        for b, plist in enumerate(self.buckets):
            for p in plist:
                self._bucket_of[p] = b
            self._flat.append(torch.empty(sum(p.numel() for p in plist), dtype=plist[0].dtype,
                                          device=plist[0].device))
        for p in self.active:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self._reset_counts()

    def _reset_counts(self):
        self._remaining = [len(b) for b in self.buckets]
        self._pending = []

    # -- per-step -------------------------------------------------------------------------------
    def _launch(self, b):
        plist, flat = self.buckets[b], self._flat[b]
        if self.use_cuda:
            self.stream.wait_stream(torch.cuda.current_stream())
            ctx = torch.cuda.stream(self.stream)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        with ctx:
            off = 0
            for p in plist:  # pack, pre-divided by the world size
                n = p.numel()
                torch.mul(p.grad.reshape(-1), 1.0 / self.world_size, out=flat[off:off + n])
                off += n
            work = dist.all_reduce(flat, group=self.group, async_op=True)
        self._pending.append((b, work))

    def _on_grad(self, p):
        b = self._bucket_of[p]
        self._remaining[b] -= 1
        if self._remaining[b] == 0:
            self._launch(b)

    def finish(self):
        """Call after ``loss.backward()``: waits for the buckets and writes averaged grads back."""
        if self.buckets is None:
            # first step: plain (reference-style) all-reduce, and learn which params get gradients
            self.active = [p for p in self.params if p.grad is not None]
            _allreduce_coalesced([p.grad.data for p in self.active], self.world_size, -1)
            self._build()
            return
        for b, r in enumerate(self._remaining):  # buckets whose hooks did not all fire this step
            if r > 0:
                self._launch(b)
        for b, work in self._pending:
            work.wait()
            if self.use_cuda:
                ctx = torch.cuda.stream(self.stream)
            else:
                import contextlib
                ctx = contextlib.nullcontext()
            with ctx:
                off = 0
                for p in self.buckets[b]:
                    n = p.numel()
                    if p.grad is not None:
                        p.grad.copy_(self._flat[b][off:off + n].view_as(p.grad))
                    off += n
        if self.use_cuda:
            torch.cuda.current_stream().wait_stream(self.stream)
        self._reset_counts()


class DistOptimizerHook(object):
    """zero_grad -> backward -> all-reduce -> clip -> step (dist_utils.py:44-58), callable without mmcv."""

    def __init__(self, grad_clip=None, coalesce=True, bucket_size_mb=-1, overlap=False):
        self.grad_clip = grad_clip
        self.coalesce = coalesce
        self.bucket_size_mb = bucket_size_mb
        self.overlap = overlap
        self._reducer = None
        self._params = None

    def clip_grads(self, params):
        return clip_grads(params, **self.grad_clip)

    def step(self, model, optimizer, loss):
        optimizer.zero_grad()
        distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if distributed and self.overlap and self._reducer is None:
            self._reducer = OverlappedGradReducer(list(model.parameters()),
                                                  self.bucket_size_mb if self.bucket_size_mb > 0 else 32)
        loss.backward()
        if distributed:
            if self._reducer is not None:
                self._reducer.finish()
            else:
                allreduce_grads(model.parameters(), self.coalesce, self.bucket_size_mb)
        if self.grad_clip is not None:
            if self._params is None:    # model.parameters() walks every module: 1 ms of Python per step
                self._params = [p for p in model.parameters() if p.requires_grad]
            self.clip_grads(self._params)
        optimizer.step()
