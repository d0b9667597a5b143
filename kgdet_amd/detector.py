"""``RepPointsDetectorKp``: backbone -> neck -> keypoint-guided head, result packing.

Mirrors mmdet/models/detectors/reppoints_detector_kp.py:9-148 on top of
single_stage.py:9-70 and base.py:12-142 (``forward(img, img_meta, return_loss=True, **kw)``
dispatch, ``forward_train``, ``simple_test``, ``bbox2result_kp``).  The reference asserts one
image per GPU at test time (base.py:75-76); ``simple_test_batch`` lifts that for the batch-8
inference configuration, running decode + NMS for the whole batch at once.
"""
import numpy as np
import torch
import torch.nn as nn

from .registry import DETECTORS, build_backbone, build_head, build_neck


_PINNED_RESULTS = __import__('os').environ.get('KGDET_PINNED_RESULTS', '1') == '1'    # 0: `.cpu()` into pageable memory (A/B)


@DETECTORS.register_module
class RepPointsDetectorKp(nn.Module):

    def __init__(self, backbone, neck, bbox_head, train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__()
        self.backbone = build_backbone(backbone)
        if neck is not None:
            self.neck = build_neck(neck)
        self.bbox_head = build_head(bbox_head)
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg
        self.init_weights(pretrained=pretrained)

    @property
    def with_neck(self):
        return hasattr(self, 'neck') and self.neck is not None

    @property
    def with_keypoint(self):
        return True

    def init_weights(self, pretrained=None):
        if isinstance(pretrained, str) and pretrained.startswith(('modelzoo://', 'open-mmlab://', 'http')):
            # there is no network: remote checkpoints cannot be fetched, weights stay randomly initialised
            pretrained = None
        self.backbone.init_weights(pretrained=pretrained)
        if self.with_neck:
            if isinstance(self.neck, nn.Sequential):
                for m in self.neck:
                    m.init_weights()
            else:
                self.neck.init_weights()
        self.bbox_head.init_weights()

    def extract_feat(self, img):
        x = self.backbone(img)
        if self.with_neck:
            x = self.neck(x)
        if not torch.is_grad_enabled() and not getattr(self.bbox_head, 'channels_last_inference', False):
            # the bf16 inference backbone runs channels-last (backbone.conv_bn); the head's kernels take NCHW
            # (a head whose towers stay channels-last -- heads_serial -- takes the maps as they come)
            x = tuple(o.contiguous() for o in x)
        return x

    def forward_dummy(self, img):
        return self.bbox_head(self.extract_feat(img), None)

    def train(self, mode=True):
        """``nn.Module.train`` + every derived inference-time copy of the weights dropped (autocast casts, folded backbone, packed
        deformable operands): they are keyed on the parameters' version counters, which a replayed HIP graph of the training
        step (``runner.GraphedTrainStep``) or a write through ``.data`` does not move.  ``eval()`` calls ``train(False)``."""
        from . import conv1x1
        conv1x1.invalidate_inference_caches()
        return super(RepPointsDetectorKp, self).train(mode)

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_keypoints, gt_bboxes_ignore=None):
        from . import conv1x1
        with conv1x1.step_scope():    # the split-bf16 convolutions' weight images: one pack launch per step
            x = self.extract_feat(img)
            outs = self.bbox_head(x, img_metas)
        loss_inputs = outs + (gt_bboxes, gt_labels, gt_keypoints, img_metas, self.train_cfg)
        return self.bbox_head.loss(*loss_inputs, gt_bboxes_ignore=gt_bboxes_ignore)

    def bbox2result_kp(self, bboxes, labels, kpts, num_classes):
        """per-class numpy lists: (bboxes_in_cls, bbox_scores, kpt_in_cls), or a 1-tuple when empty"""
        if bboxes.shape[0] == 0:
            return ([np.zeros((0, 5), dtype=np.float32) for i in range(num_classes - 1)], )
        if torch.is_tensor(bboxes):
            bboxes, labels, kpts = bboxes.float().cpu().numpy(), labels.cpu().numpy(), kpts.float().cpu().numpy()
        return ([bboxes[labels == i, :] for i in range(num_classes - 1)], bboxes[:, 4],
                [kpts[labels == i, :] for i in range(num_classes - 1)])

    def simple_test_batch(self, img, img_meta, rescale=False):
        x = self.extract_feat(img)
        outs = self.bbox_head(x, img_meta)
        get = getattr(self.bbox_head, 'get_bboxes_numpy', self.bbox_head.get_bboxes)   # one D2H copy per batch
        bbox_list = get(*(outs + (img_meta, self.test_cfg, rescale)))
        return [self.bbox2result_kp(det_bboxes, det_labels, det_kpts, self.bbox_head.num_classes)
                for det_bboxes, det_labels, det_kpts in bbox_list]

    def graphed_test_batch(self, img, img_meta, rescale=False, autocast_dtype=None, warmup=3):
        """``simple_test_batch`` for a fixed input shape and fixed image metas as ONE HIP-graph launch.

        A batch is ~360 kernel launches whose CPU-side issue (Python module calls + launch latency) takes about as
        long as the GPU needs to run them; backbone, neck, head, decode and the fused NMS have no host read, so the
        whole chain is captured once (``torch.cuda.CUDAGraph`` = hipGraph) and replayed per batch, followed by the
        single device->host copy of the packed results.  Returns ``run(img) -> results`` (same results as
        ``simple_test_batch``); ``run.static_img`` is the graph's input buffer -- ``run(run.static_img)`` after writing the
        batch into it (e.g. as the destination of the loader's host->device copy) skips the 103 MB device copy; raises NotImplementedError when the head's packed post-processing does not apply."""
        import contextlib
        assert img.is_cuda and not self.training
        scope = (lambda: torch.autocast('cuda', dtype=autocast_dtype)) if autocast_dtype is not None \
            else contextlib.nullcontext
        static_img = img.clone()

        def chain():
            outs = self.bbox_head(self.extract_feat(static_img), img_meta)
            return self.bbox_head.get_bboxes_packed_tensor(*(outs + (img_meta, self.test_cfg, rescale)))

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad(), scope():
            for _ in range(warmup):          # MIOpen find, lazy module loads, weight caches: all before the capture
                packed = chain()
        torch.cuda.current_stream().wait_stream(side)
        if packed is None:
            raise NotImplementedError('packed post-processing not applicable to this head / test_cfg / metas')
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), scope(), torch.cuda.graph(graph):
            static_out = chain()
        num_classes = self.bbox_head.num_classes

        # the results land in a page-locked buffer (a `.cpu()` into pageable memory is staged by the driver: 232 us for the
        # 2.8 MB of a batch of 8 against ~60); bbox2result_kp copies the valid rows out of it, so it is reused per batch
        host_out = torch.empty(static_out.shape, dtype=static_out.dtype, pin_memory=True)

        def run(new_img):
            if new_img is not static_img:       # (run.static_img IS the input buffer: a loader that writes into it saves the copy)
                static_img.copy_(new_img)
            graph.replay()
            if _PINNED_RESULTS:
                host_out.copy_(static_out, non_blocking=True)
                torch.cuda.current_stream().synchronize()
                dets = self.bbox_head.unpack_results(host_out.numpy())
            else:
                dets = self.bbox_head.unpack_results(static_out.cpu().numpy())
            # (d.copy(): the score column of the result is a view of d; labels and the per-class rows are copies already)
            return [self.bbox2result_kp(d.copy(), lab, k, num_classes) for d, lab, k in dets]

        # Pipelined use (a serving loop): ``slot = run.submit()`` replays the graph on the batch in ``run.static_img`` and starts
        # the device->host copy of its packed results on a side stream; ``run.collect(slot)`` waits for that copy and unpacks.
        # Calling submit(k + 1) before collect(k) puts batch k's copy and host-side unpacking UNDER batch k + 1's kernels (the
        # serial ``run`` leaves the GPU idle for both).  Two result slots: collect(k) must have returned before submit(k + 2).
        stash = [torch.empty_like(static_out) for _ in range(2)]
        hosts = [torch.empty(static_out.shape, dtype=static_out.dtype, pin_memory=True) for _ in range(2)]
        ready = [torch.cuda.Event() for _ in range(2)]
        landed = [torch.cuda.Event() for _ in range(2)]
        copy_stream = torch.cuda.Stream()
        counter = [0]

        def submit(new_img=None):
            if new_img is not None and new_img is not static_img:
                static_img.copy_(new_img)
            slot = counter[0] & 1
            counter[0] += 1
            graph.replay()
            stash[slot].copy_(static_out)              # (2.8 MB device copy: the next replay overwrites static_out)
            ready[slot].record()
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(ready[slot])
                hosts[slot].copy_(stash[slot], non_blocking=True)
                landed[slot].record()
            return slot

        def collect(slot):
            landed[slot].synchronize()
            dets = self.bbox_head.unpack_results(hosts[slot].numpy())
            return [self.bbox2result_kp(d.copy(), lab, k, num_classes) for d, lab, k in dets]

        run.submit, run.collect = submit, collect
        run.graph, run.static_img, run.static_out = graph, static_img, static_out
        # The captured kernels hold raw pointers into the module-level weight-image caches (packed deformable-conv
        # operands, folded conv+BN weights).  Those caches are cleared on mode switches / when they fill up, which
        # would free memory the graph still reads: the graph keeps its own strong references.
        from . import backbone, dcn
        run.pinned = ([v[2] for v in dcn._pack_cache.values()], [v[1:4] for v in backbone._fold_cache.values()])
        return run

    def simple_test(self, img, img_meta, rescale=False):
        return self.simple_test_batch(img, img_meta, rescale)[0]

    def aug_test(self, imgs, img_metas, rescale=False):
        raise NotImplementedError

    def forward_test(self, imgs, img_metas, **kwargs):
        for var, name in [(imgs, 'imgs'), (img_metas, 'img_metas')]:
            if not isinstance(var, list):
                raise TypeError('{} must be a list, but got {}'.format(name, type(var)))
        num_augs = len(imgs)
        if num_augs != len(img_metas):
            raise ValueError('num of augmentations ({}) != num of image meta ({})'.format(len(imgs), len(img_metas)))
        imgs_per_gpu = imgs[0].size(0)
        assert imgs_per_gpu == 1
        if num_augs == 1:
            return self.simple_test(imgs[0], img_metas[0], **kwargs)
        return self.aug_test(imgs, img_metas, **kwargs)

    def forward(self, img, img_meta, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(img, img_meta, **kwargs)
        return self.forward_test(img, img_meta, **kwargs)
