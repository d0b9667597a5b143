"""Training runtime of the KGDet configs without mmcv  (SURVEY 8f row 3).

What the reference gets from ``mmcv==0.2.13``'s ``Runner`` + hooks (third-party, not in the reference tree; pinned by
``requirements.txt:7``) and ``mmdet/apis/train.py``, restated as one small loop:

* ``parse_losses`` / ``batch_processor`` / ``build_optimizer``  -- ``mmdet/apis/train.py:17-134``.
* ``LrSchedule`` -- mmcv's ``LrUpdaterHook`` for ``lr_config = dict(policy='step', warmup='linear', warmup_iters=500,
  warmup_ratio=1/3, step=[8, 11])`` (``configs/kgdet_moment_r50_fpn_1x-demo.py:134-139``): the regular rate is
  ``base_lr * gamma ** (#steps already passed)`` set at the start of every epoch; during the first ``warmup_iters``
  iterations it is scaled by ``1 - (1 - it / warmup_iters) * (1 - warmup_ratio)`` ('linear'), ``warmup_ratio``
  ('constant') or ``warmup_ratio ** (1 - it / warmup_iters)`` ('exp'); at ``it == warmup_iters`` it returns to the
  regular rate.
* ``Runner`` -- epoch loop, ``DistOptimizerHook`` (all-reduce + grad-clip 35 + step), checkpoint every ``interval``
  epochs as ``epoch_{n}.pth`` + ``latest.pth`` with ``meta = {epoch, iter}`` and the optimizer state, ``resume``.
* ``single_gpu_test`` / ``multi_gpu_test`` / ``collect_results`` -- ``mmdetection/tools/test.py:18-100``: inference over a
  dataset, sharded by rank (sample i of the padded index list goes to rank i % world_size, the ``DistributedSampler(shuffle=False)``
  of ``mmdet/datasets/loader/build_loader.py``), results gathered on rank 0 in dataset order, padding samples dropped.
"""
import os
import shutil
import time
from collections import OrderedDict

import torch

from .checkpoint import load_checkpoint, save_checkpoint
from .dist import DistOptimizerHook


def parse_losses(losses):
    log_vars = OrderedDict()
    for name, value in losses.items():
        if isinstance(value, torch.Tensor):
            log_vars[name] = value.mean()
        elif isinstance(value, list):
            log_vars[name] = sum(v.mean() for v in value)
        else:
            raise TypeError('{} is not a tensor or list of tensors'.format(name))
    loss = sum(v for k, v in log_vars.items() if 'loss' in k)
    log_vars['loss'] = loss
    return loss, log_vars


def batch_processor(model, data, train_mode=True):
    losses = model(**data)
    loss, log_vars = parse_losses(losses)
    return dict(loss=loss, log_vars=log_vars, num_samples=len(data['img']))


def build_optimizer(model, optimizer_cfg):
    import re
    model = model.module if hasattr(model, 'module') else model
    cfg = dict(optimizer_cfg)
    paramwise = cfg.pop('paramwise_options', None)
    cls = getattr(torch.optim, cfg.pop('type'))
    if paramwise is None:
        return cls(model.parameters(), **cfg)
    base_lr, base_wd = cfg['lr'], cfg.get('weight_decay', None)
    if 'bias_decay_mult' in paramwise or 'norm_decay_mult' in paramwise:
        assert base_wd is not None
    groups = []
    for name, param in model.named_parameters():
        group = {'params': [param]}
        if param.requires_grad:
            if re.search(r'(bn|gn)(\d+)?.(weight|bias)', name):
                if base_wd is not None:
                    group['weight_decay'] = base_wd * paramwise.get('norm_decay_mult', 1.)
            elif name.endswith('.bias'):
                group['lr'] = base_lr * paramwise.get('bias_lr_mult', 1.)
                if base_wd is not None:
                    group['weight_decay'] = base_wd * paramwise.get('bias_decay_mult', 1.)
        groups.append(group)
    return cls(groups, **cfg)


class LrSchedule(object):
    def __init__(self, policy='step', step=(), gamma=0.1, by_epoch=True, warmup=None, warmup_iters=0,
                 warmup_ratio=0.1, **unused):
        if policy != 'step':
            raise NotImplementedError('lr policy {!r} (the KGDet configs use "step")'.format(policy))
        if warmup is not None:
            if warmup not in ('constant', 'linear', 'exp'):
                raise ValueError('"{}" is not a supported type for warming up'.format(warmup))
            assert warmup_iters > 0 and 0 < warmup_ratio <= 1.0
        self.step = step if isinstance(step, int) else list(step)
        assert (self.step > 0) if isinstance(self.step, int) else all(s > 0 for s in self.step)
        self.gamma, self.by_epoch = gamma, by_epoch
        self.warmup, self.warmup_iters, self.warmup_ratio = warmup, warmup_iters, warmup_ratio
        self.base_lr, self.regular_lr = [], []

    def before_run(self, optimizer):
        for g in optimizer.param_groups:
            g.setdefault('initial_lr', g['lr'])
        self.base_lr = [g['initial_lr'] for g in optimizer.param_groups]

    def regular(self, progress):
        if isinstance(self.step, int):
            return [lr * self.gamma ** (progress // self.step) for lr in self.base_lr]
        passed = len(self.step)
        for i, s in enumerate(self.step):
            if progress < s:
                passed = i
                break
        return [lr * self.gamma ** passed for lr in self.base_lr]

    def warm(self, cur_iter):
        if self.warmup == 'constant':
            k = self.warmup_ratio
        elif self.warmup == 'linear':
            k = 1 - (1 - cur_iter / self.warmup_iters) * (1 - self.warmup_ratio)
        else:
            k = self.warmup_ratio ** (1 - cur_iter / self.warmup_iters)
        return [lr * k for lr in self.regular_lr]

    @staticmethod
    def _set(optimizer, lrs):
        for g, lr in zip(optimizer.param_groups, lrs):
            g['lr'] = lr

    def before_train_epoch(self, optimizer, epoch):
        if self.by_epoch:
            self.regular_lr = self.regular(epoch)
            self._set(optimizer, self.regular_lr)

    def before_train_iter(self, optimizer, cur_iter):
        if not self.by_epoch:
            self.regular_lr = self.regular(cur_iter)
            if self.warmup is None or cur_iter >= self.warmup_iters:
                self._set(optimizer, self.regular_lr)
            else:
                self._set(optimizer, self.warm(cur_iter))
        elif self.warmup is not None and cur_iter <= self.warmup_iters:
            self._set(optimizer, self.regular_lr if cur_iter == self.warmup_iters else self.warm(cur_iter))


def test_shard(n, world_size, rank):
    """Indices of rank ``rank``: the reference's test-time ``DistributedSampler(dataset, world_size, rank, shuffle=False)`` -- the
    index list padded by wrapping to a multiple of ``world_size``, then every ``world_size``-th index from ``rank`` on."""
    total = int(-(-n // world_size)) * world_size
    idx = list(range(n))
    if n > 0:
        idx = (idx * (-(-total // n)))[:total]      # (wrapping as often as needed: n < world_size / 2 leaves no rank empty)
    assert len(idx) == total or n == 0
    return idx[rank:total:world_size]


def _test_on(model, dataset, indices, rescale, to_device, imgs_per_gpu=1):
    """results of the samples ``indices`` in that order.  imgs_per_gpu == 1: the reference's call, one image per forward
    (``model(return_loss=False, rescale=..., **data)``, tools/test.py:25-28); > 1: runs of consecutive samples with identical
    tensor shapes and one augmentation go through ``simple_test_batch`` together (the same per-image results: nothing in
    backbone / neck / head / decode / NMS mixes the images of a batch)."""
    model.eval()
    results, i = [], 0
    carried = None                 # the sample that ended the previous group (loaded once)
    while i < len(indices):
        data = carried if carried is not None else dataset[indices[i]]
        carried = None
        group = [data]
        while (imgs_per_gpu > 1 and len(group) < imgs_per_gpu and i + len(group) < len(indices) and len(data['img']) == 1):
            nxt = dataset[indices[i + len(group)]]
            if len(nxt['img']) != 1 or nxt['img'][0].shape != data['img'][0].shape:
                carried = nxt
                break
            group.append(nxt)
        with torch.no_grad():
            if len(group) == 1:
                imgs = [t[None] for t in data['img']]
                metas = [[m] for m in data['img_meta']]
                if to_device is not None:
                    imgs = [to_device(t) for t in imgs]
                results.append(model(imgs, metas, return_loss=False, rescale=rescale))
            else:
                img = torch.stack([g['img'][0] for g in group])
                if to_device is not None:
                    img = to_device(img)
                results.extend(model.simple_test_batch(img, [g['img_meta'][0] for g in group], rescale=rescale))
        i += len(group)
    return results


def single_gpu_test(model, dataset, rescale=True, to_device=None, imgs_per_gpu=1):
    """tools/test.py:18-35 without the progress bar: one result per sample, in dataset order"""
    return _test_on(model, dataset, list(range(len(dataset))), rescale, to_device, imgs_per_gpu)


def collect_results(result_part, size, group=None):
    """tools/test.py:61-100 with the gather done by the process group instead of pickles in a shared temporary directory:
    rank 0 receives every rank's list, interleaves them (sample k of rank r is dataset index k * world_size + r), cuts the
    padding samples off and returns ``size`` results; the other ranks return None."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return list(result_part)[:size]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    parts = [None] * world if rank == 0 else None
    dist.gather_object(result_part, parts, dst=0, group=group)
    if rank != 0:
        return None
    if len(set(len(p) for p in parts)) != 1:
        raise RuntimeError('ranks returned different numbers of results: %s (the shards of test_shard are equally long)'
                           % [len(p) for p in parts])
    ordered = []
    for res in zip(*parts):          # (every rank holds the same number of samples: the index list was padded)
        ordered.extend(list(res))
    return ordered[:size]


def multi_gpu_test(model, dataset, rescale=True, to_device=None, imgs_per_gpu=1, group=None):
    """tools/test.py:38-58: every rank runs its shard of the dataset, rank 0 returns all results in dataset order"""
    import torch.distributed as dist
    on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if on else 1
    rank = dist.get_rank(group) if on else 0
    part = _test_on(model, dataset, test_shard(len(dataset), world, rank), rescale, to_device, imgs_per_gpu)
    return collect_results(part, len(dataset), group)


class GraphedTrainStep(object):
    """One training step of a FIXED-shape batch -- forward, the nine losses, backward, gradient clip, Adam (or torch's fused
    SGD: config 5) -- as ONE HIP graph.

    The eager step is ~560 kernel launches whose enqueue (Python module calls + launch latency, ~11 ms at full size) takes as
    long as the GPU needs to run them (``tools/host_step_cost.py``): every kernel gain below a few percent is invisible in
    images/s.  Nothing in the step reads back to the host (``tests/test_gpu_head.py::test_training_step_has_no_host_syncs``), the
    optimizer's schedule lives in device memory (``optim.FusedClipAdam.enable_device_schedule``: step count, bias corrections
    and the learning rate are not kernel arguments), so the whole step is captured once (``torch.cuda.CUDAGraph`` = hipGraph)
    and replayed per iteration.  The reference has no equivalent (``mmdet/apis/train.py:17-134`` + mmcv's Runner issue every
    op eagerly); the arithmetic is the eager step's, kernel by kernel.

    ``batch``: the dict ``forward_train`` takes (``img`` [B, 3, H, W], ``img_meta``, ``gt_bboxes`` / ``gt_labels`` /
    ``gt_keypoints`` lists of CUDA tensors): these tensors become the graph's input buffers; ``load(batch)`` copies another
    batch of the SAME shapes (same number of ground-truth rows per image) into them, ``step()`` publishes the step's learning
    rate (``optimizer.param_groups[0]['lr']``, as set by ``LrSchedule``), replays the graph and returns the loss tensors of
    the step (device tensors, overwritten by the next replay).  One rank only: the gradient exchange of a multi-rank job
    stays on the eager path (``DistOptimizerHook``).  ``sync_optimizer_state()`` before the optimizer's state is read
    (checkpoints): the replayed steps are added to its ``step`` counters then."""

    def __init__(self, model, optimizer, opt_hook, batch, warmup=3, batch_processor=batch_processor):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            raise NotImplementedError('the graphed step covers one rank; a multi-rank job exchanges gradients eagerly')
        from .optim import FusedClipAdam
        self.model, self.optimizer, self.hook = model, optimizer, opt_hook
        self.static = batch
        self.static_tensors = [batch['img']] + [t for k in ('gt_bboxes', 'gt_labels', 'gt_keypoints') for t in batch.get(k, [])]
        assert all(t.is_cuda for t in self.static_tensors) and model.training

        def one_step():
            if getattr(self, 'lr_t', None) is not None and torch.cuda.is_current_stream_capturing():
                self.lr_t.copy_(self.lr_pin[0], non_blocking=True)
            out = batch_processor(model, self.static, train_mode=True)
            opt_hook.step(model, optimizer, out['loss'])
            return out

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(int(warmup), 2)):      # optimizer state, weight-image sets, MIOpen find: all before the capture
                one_step()
        torch.cuda.current_stream().wait_stream(side)
        fused = opt_hook._fused
        self.fused, self.lr_t = None, None
        if fused is not None and len(optimizer.param_groups) == 1 and \
                FusedClipAdam.applicable(optimizer, opt_hook._params, opt_hook.grad_clip):
            if fused._sched is None:
                fused.enable_device_schedule(optimizer)
            self.fused = fused
        elif (type(optimizer) is torch.optim.SGD and len(optimizer.param_groups) == 1
              and optimizer.param_groups[0].get('fused') and not optimizer.param_groups[0].get('nesterov')):
            # torch's fused SGD (config 5: momentum 0.9, weight decay 1e-4) keeps no step count; with the learning rate as a
            # DEVICE scalar the whole step -- clip_grad_norm_'s foreach chain included -- is capturable: `step()` writes the
            # scheduler's rate into that scalar before every replay (a fill kernel, no read-back)
            group = optimizer.param_groups[0]
            self.lr_t = torch.tensor(float(group['lr']), dtype=torch.float32, device=self.static_tensors[0].device)
            # the rate reaches the device scalar through a copy node at the head of the graph, from ONE page-locked host scalar that
            # `step()` rewrites only when the scheduler's value changed (after waiting for the replays that still read the old one)
            self.lr_pin = torch.tensor([float(group['lr'])], dtype=torch.float32).pin_memory()
            self._lr_last = float(group['lr'])
        else:
            raise NotImplementedError('the graphed step needs the fused clip + Adam step or torch.optim.SGD(fused=True) '
                                      '(one parameter group)')
        group = optimizer.param_groups[0]
        lr_host = group['lr']
        if self.lr_t is not None:
            group['lr'] = self.lr_t
        try:
            with torch.cuda.stream(side):                  # one eager step in the captured form (its table, its buffers)
                one_step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            optimizer.zero_grad(set_to_none=True)
            if self.fused is not None:
                self.fused.publish_lr(optimizer)
            with torch.cuda.graph(self.graph):
                self.out = one_step()
        finally:
            if self.lr_t is not None:
                group['lr'] = lr_host                      # (schedulers keep seeing and setting a float)
        # (the capture itself does not run the step: nothing was counted yet)
        self.steps = 0

    def load(self, batch):
        new = [batch['img']] + [t for k in ('gt_bboxes', 'gt_labels', 'gt_keypoints') for t in batch.get(k, [])]
        if len(new) != len(self.static_tensors) or any(a.shape != b.shape for a, b in zip(new, self.static_tensors)):
            raise ValueError('the graphed step was captured for other tensor shapes')
        if self.lr_t is not None:
            # Observed on this stack (ROCm 7.0 / torch 2.10, tools/graph_serial_probe.py): an eager kernel enqueued BEHIND an in-flight
            # replay of the config-5 (serial head, SGD) graph ends in an HSA hardware exception (0x1016) -- the KGDet graph takes the
            # same pattern without complaint (tools/graph_load_probe.py), a replay that has finished is fine in both.  Cause not
            # found; until it is, this flow waits for the replay before it touches the graph's input buffers.
            torch.cuda.current_stream().synchronize()
        with torch.no_grad():
            torch._foreach_copy_(self.static_tensors, [t.to(s.device, non_blocking=True) for t, s in zip(new, self.static_tensors)])

    def step(self):
        if self.fused is not None:
            self.fused.publish_lr(self.optimizer)
            self.graph.replay()
            self.fused.step_published()
        else:
            lr = float(self.optimizer.param_groups[0]['lr'])
            if lr != self._lr_last:
                torch.cuda.current_stream().synchronize()     # (replays in flight read the scalar's old value)
                self.lr_pin[0] = lr
                self._lr_last = lr
            self.graph.replay()
        self.steps += 1
        # the replayed kernels wrote the parameters through raw pointers: the version counters did not move, and the inference-time
        # copies derived from the weights (conv1x1._cast_cache, the folded backbone, the packed deformable operands) are keyed on them
        self._stale_caches = True
        return self.out

    def refresh_inference_caches(self):
        """after replays, before an evaluation pass of the same model: drops every derived copy of the stepped weights (also done
        by ``model.eval()`` / ``model.train()`` of the detector, detector.py)"""
        if getattr(self, '_stale_caches', False):
            from . import conv1x1
            conv1x1.invalidate_inference_caches()
            self._stale_caches = False

    def sync_optimizer_state(self):
        if self.fused is not None:
            self.fused.sync_optimizer_state(self.optimizer)


class Runner(object):
    def __init__(self, model, optimizer, work_dir=None, lr_config=None, optimizer_config=None, checkpoint_config=None,
                 log_interval=50, logger=print, batch_processor=batch_processor):
        self.model, self.optimizer, self.work_dir = model, optimizer, work_dir
        self.lr = LrSchedule(**(lr_config or dict(policy='step', step=[])))
        self.opt_hook = DistOptimizerHook(**(optimizer_config or {}))
        self.ckpt_interval = (checkpoint_config or {}).get('interval', 1)
        self.log_interval, self.logger, self.batch_processor = log_interval, logger, batch_processor
        self.epoch, self.iter = 0, 0
        self.log_history = []

    def current_lr(self):
        return [g['lr'] for g in self.optimizer.param_groups]

    def save_checkpoint(self, filename_tmpl='epoch_{}.pth', create_latest=True):
        path = os.path.join(self.work_dir, filename_tmpl.format(self.epoch))
        save_checkpoint(self.model, path, optimizer=self.optimizer, meta=dict(epoch=self.epoch, iter=self.iter))
        if create_latest:
            shutil.copyfile(path, os.path.join(self.work_dir, 'latest.pth'))
        return path

    def resume(self, checkpoint, resume_optimizer=True, map_location='cpu'):
        ckpt = load_checkpoint(self.model, checkpoint, map_location=map_location)
        self.epoch, self.iter = ckpt['meta']['epoch'], ckpt['meta']['iter']
        if 'optimizer' in ckpt and resume_optimizer:
            self.optimizer.load_state_dict(ckpt['optimizer'])
        self.logger('resumed epoch %d, iter %d' % (self.epoch, self.iter))
        return ckpt

    def train_epoch(self, data_loader, to_device=None):
        self.model.train()
        self.lr.before_train_epoch(self.optimizer, self.epoch)
        t0 = time.time()
        for i, data in enumerate(data_loader):
            self.lr.before_train_iter(self.optimizer, self.iter)
            if to_device is not None:
                data = to_device(data)
            out = self.batch_processor(self.model, data, train_mode=True)
            self.opt_hook.step(self.model, self.optimizer, out['loss'])
            self.iter += 1
            if self.log_interval and (i + 1) % self.log_interval == 0:
                rec = OrderedDict(epoch=self.epoch + 1, iter=i + 1, lr=self.current_lr()[0],
                                  time=(time.time() - t0) / (i + 1))
                rec.update((k, float(v.detach()) if isinstance(v, torch.Tensor) else float(v)) for k, v in out['log_vars'].items())     # the only device->host read
                self.log_history.append(rec)
                self.logger(', '.join('%s: %.5g' % kv if isinstance(kv[1], float) else '%s: %s' % kv for kv in rec.items()))
        self.epoch += 1

    def run(self, data_loader, max_epochs, to_device=None, set_epoch=None):
        self.lr.before_run(self.optimizer)
        while self.epoch < max_epochs:
            if set_epoch is not None:
                set_epoch(self.epoch)
            self.train_epoch(data_loader, to_device)
            if self.work_dir is not None and self.ckpt_interval > 0 and self.epoch % self.ckpt_interval == 0:
                self.save_checkpoint()
        return self
