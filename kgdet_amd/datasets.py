"""DeepFashion2 input pipeline without mmcv / cv2  (data side).

* ``ImageTransform`` / ``bbox_transform`` / ``keypoint_transform`` -- ``mmdet/datasets/transforms.py:11-169``:
  keep-ratio rescale to ``img_scale`` (``mmcv.imrescale``: factor = min(long/max(h,w), short/min(h,w)), new size
  ``int(x * f + 0.5)``), normalise, flip, zero-pad bottom/right to a multiple of ``size_divisor``, CHW; boxes are
  scaled, flipped (``w - x - 1``), clipped to ``img_shape``; landmarks are scaled, flipped and their left/right
  partners exchanged per category.
* ``DeepFashion2Dataset`` -- ``mmdet/datasets/deepfashion2.py:8-113`` + ``coco.py:29-160`` + ``custom.py:56-381``:
  same constructor keywords as the configs pass, annotation filtering (``_filter_imgs``), aspect-ratio ``flag``,
  ``prepare_train_img`` / ``prepare_test_img`` outputs (plain tensors instead of mmcv DataContainers).
* ``GroupSampler`` / ``DistributedGroupSampler`` -- ``mmdet/datasets/loader/sampler.py:38-164``; ``collate`` pads a
  batch's images to a common size like mmcv's ``collate`` does for stacked DataContainers.

Decoding uses PIL (RGB) where the reference uses cv2 (BGR + ``to_rgb``); the resize is bilinear with half-pixel
centres and no antialiasing (cv2.INTER_LINEAR geometry) evaluated in float and rounded to uint8, which can differ
from cv2's 11-bit fixed-point result by one grey level.  The reference cannot be run here (no mmcv / cv2), so this
part is pinned by closed-form cases in tests/test_datasets.py, not by reference outputs.
"""
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

from .evaluation import CocoIndex, landmark_meta
from .registry import DATASETS


def rescale_size(h, w, scale):
    """``mmcv.imrescale`` size rule for a (long, short) bound -> (new_h, new_w, factor)."""
    long_edge, short_edge = max(scale), min(scale)
    f = min(long_edge / max(h, w), short_edge / min(h, w))
    return int(h * float(f) + 0.5), int(w * float(f) + 0.5), f


def resize_bilinear_u8(img, new_h, new_w):
    """uint8 HxWx3 -> uint8 new_h x new_w x 3, bilinear, half-pixel centres, edge-clamped, no antialias."""
    t = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1)[None].float()
    t = F.interpolate(t, size=(new_h, new_w), mode='bilinear', align_corners=False)
    return t[0].permute(1, 2, 0).round_().clamp_(0, 255).to(torch.uint8).numpy()


class ImageTransform(object):
    def __init__(self, mean=(0, 0, 0), std=(1, 1, 1), to_rgb=True, size_divisor=None):
        self.mean = np.array(mean, dtype=np.float32)
        self.std = np.array(std, dtype=np.float32)
        self.to_rgb = to_rgb          # images are decoded as RGB here; False = the statistics are in BGR order
        self.size_divisor = size_divisor

    def __call__(self, img, scale, flip=False, keep_ratio=True):
        h, w = img.shape[:2]
        if keep_ratio:
            new_h, new_w, scale_factor = rescale_size(h, w, scale)
        else:
            new_w, new_h = scale
            scale_factor = np.array([new_w / w, new_h / h, new_w / w, new_h / h], dtype=np.float32)
        img = resize_bilinear_u8(img, new_h, new_w)
        img_shape = img.shape
        img = img.astype(np.float32)
        if not self.to_rgb:
            img = img[..., ::-1]
        img = (img - self.mean) / self.std
        if flip:
            img = img[:, ::-1]
        if self.size_divisor is not None:
            ph = int(math.ceil(img.shape[0] / self.size_divisor)) * self.size_divisor
            pw = int(math.ceil(img.shape[1] / self.size_divisor)) * self.size_divisor
            padded = np.zeros((ph, pw, 3), dtype=np.float32)
            padded[:img.shape[0], :img.shape[1]] = img
            img = padded
        pad_shape = img.shape
        return np.ascontiguousarray(img.transpose(2, 0, 1)), img_shape, pad_shape, scale_factor


def bbox_transform(bboxes, img_shape, scale_factor, flip=False):
    out = bboxes * scale_factor
    if flip:
        w = img_shape[1]
        flipped = out.copy()
        flipped[..., 0::4] = w - out[..., 2::4] - 1
        flipped[..., 2::4] = w - out[..., 0::4] - 1
        out = flipped
    out[:, 0::2] = np.clip(out[:, 0::2], 0, img_shape[1] - 1)
    out[:, 1::2] = np.clip(out[:, 1::2], 0, img_shape[0] - 1)
    return out


def keypoint_transform(keypoints, img_shape, gt_labels, scale_factor, swap_pairs, flip=False):
    """list of [294, 3] (x, y, v) -> [G, 294, 3]; a flip mirrors x and exchanges the category's left/right pairs."""
    out = []
    for label, kp in zip(gt_labels, keypoints):
        kp = np.c_[kp[:, 0:2] * scale_factor, kp[:, 2]]
        if flip:
            kp[:, 0] = img_shape[1] - kp[:, 0] - 1
            for a, b in swap_pairs[label - 1]:
                kp[[a, b]] = kp[[b, a]]
        out.append(kp)
    return np.stack(out, axis=0)


@DATASETS.register_module
class DeepFashion2Dataset(object):
    CLASSES = tuple(landmark_meta()['classes'])

    def __init__(self, ann_file, img_prefix, img_scale, img_norm_cfg, multiscale_mode='value', size_divisor=None,
                 proposal_file=None, num_max_proposals=1000, flip_ratio=0, with_mask=False, with_crowd=True,
                 with_label=True, with_semantic_seg=False, with_keypoint=True, seg_prefix=None, seg_scale_factor=1,
                 extra_aug=None, resize_keep_ratio=True, corruption=None, corruption_severity=1,
                 skip_img_without_anno=True, test_mode=False, group_mode=False):
        if with_mask or with_semantic_seg or proposal_file is not None or extra_aug is not None or corruption:
            raise NotImplementedError('masks / proposals / extra augmentation are outside the KGDet path')
        meta = landmark_meta()
        self.gt_class_keypoints_dict = {c + 1: tuple(r) for c, r in enumerate(meta['landmark_ranges'])}
        self.flip_pairs = meta['swap_pairs']
        self.keypoint_groups = meta['groups']
        perm = np.arange(294)
        for pairs in self.flip_pairs:
            for a, b in pairs:
                perm[a], perm[b] = b, a
        self.flip_indices = np.stack([perm * 2, perm * 2 + 1], axis=1).reshape(-1)   # over interleaved (x, y) channels

        self.img_prefix = img_prefix
        self.coco = CocoIndex(ann_file)
        self.cat_ids = self.coco.get_cat_ids()
        self.cat2label = {cat_id: i + 1 for i, cat_id in enumerate(self.cat_ids)}
        self.img_ids = self.coco.get_img_ids()
        self.img_infos = []
        for i in self.img_ids:
            info = self.coco.load_imgs([i])[0]
            info['filename'] = info['file_name']
            self.img_infos.append(info)
        if not test_mode:
            keep = self._filter_imgs()
            self.img_infos = [self.img_infos[i] for i in keep]

        self.img_scales = img_scale if isinstance(img_scale, list) else [img_scale]
        assert all(isinstance(s, tuple) for s in self.img_scales)
        assert multiscale_mode in ('value', 'range')
        assert 0 <= flip_ratio <= 1
        self.img_norm_cfg, self.multiscale_mode = img_norm_cfg, multiscale_mode
        self.flip_ratio, self.size_divisor = flip_ratio, size_divisor
        self.with_crowd, self.with_label, self.with_keypoint = with_crowd, with_label, with_keypoint
        self.test_mode, self.group_mode = test_mode, group_mode
        self.resize_keep_ratio, self.skip_img_without_anno = resize_keep_ratio, skip_img_without_anno
        self.img_transform = ImageTransform(size_divisor=size_divisor, **img_norm_cfg)
        if not test_mode:
            self.flag = np.array([1 if info['width'] / info['height'] > 1 else 0 for info in self.img_infos],
                                 dtype=np.uint8)

    def __len__(self):
        return len(self.img_infos)

    def _filter_imgs(self, min_size=32, min_keypoint=0):
        keep_anns = [a for a in self.coco.dataset['annotations']
                     if (np.array(a['keypoints'][2::3]) > 0).sum() >= min_keypoint]
        self.coco.dataset['annotations'] = keep_anns
        self.coco.create_index()
        with_ann = set(a['image_id'] for a in keep_anns)
        return [i for i, info in enumerate(self.img_infos)
                if self.img_ids[i] in with_ann and min(info['width'], info['height']) >= min_size]

    def get_ann_info(self, idx):
        anns = self.coco.load_anns(self.coco.get_ann_ids(img_ids=[self.img_infos[idx]['id']]))
        boxes, labels, ignore, kps = [], [], [], []
        for ann in anns:
            if ann.get('ignore', False):
                continue
            x1, y1, w, h = ann['bbox']
            if ann['area'] <= 0 or w < 1 or h < 1:
                continue
            box = [x1, y1, x1 + w - 1, y1 + h - 1]
            if ann['iscrowd']:
                ignore.append(box)
            else:
                boxes.append(box)
                labels.append(self.cat2label[ann['category_id']])
            if self.with_keypoint:   # appended for crowd boxes too, as the reference does (coco.py:142-144)
                kps.append(np.reshape(ann['keypoints'], (-1, 3)))
        out = dict(bboxes=np.array(boxes, dtype=np.float32).reshape(-1, 4), labels=np.array(labels, dtype=np.int64),
                   bboxes_ignore=np.array(ignore, dtype=np.float32).reshape(-1, 4))
        if self.with_keypoint:
            out['keypoints'] = kps
        return out

    def load_image(self, idx):
        from PIL import Image
        with Image.open(os.path.join(self.img_prefix, self.img_infos[idx]['filename'])) as im:
            return np.array(im.convert('RGB'))

    def _sample_scale(self):
        scales, n = self.img_scales, len(self.img_scales)
        if n == 1:
            return scales[0]
        if n == 2 and self.multiscale_mode == 'range':
            longs, shorts = [max(s) for s in scales], [min(s) for s in scales]
            return (np.random.randint(min(longs), max(longs) + 1), np.random.randint(min(shorts), max(shorts) + 1))
        if self.multiscale_mode != 'value':
            raise ValueError('Only "value" mode supports more than 2 image scales')
        return scales[np.random.randint(n)]

    def _meta(self, info, img_shape, pad_shape, scale_factor, flip):
        return dict(ori_shape=(info['height'], info['width'], 3), img_shape=img_shape, pad_shape=pad_shape,
                    scale_factor=scale_factor, flip=flip, gt_class_keypoints_dict=self.gt_class_keypoints_dict,
                    flip_indices=self.flip_indices)

    def prepare_train_img(self, idx, img=None):
        info = self.img_infos[idx]
        img = self.load_image(idx) if img is None else img
        ann = self.get_ann_info(idx)
        if len(ann['bboxes']) == 0 and self.skip_img_without_anno:
            return None
        flip = bool(np.random.rand() < self.flip_ratio)
        scale = self._sample_scale()
        img, img_shape, pad_shape, sf = self.img_transform(img, scale, flip, keep_ratio=self.resize_keep_ratio)
        data = dict(img=torch.from_numpy(img), img_meta=self._meta(info, img_shape, pad_shape, sf, flip),
                    gt_bboxes=torch.from_numpy(bbox_transform(ann['bboxes'], img_shape, sf, flip)))
        if self.with_label:
            data['gt_labels'] = torch.from_numpy(ann['labels'])
        if self.with_crowd:
            data['gt_bboxes_ignore'] = torch.from_numpy(bbox_transform(ann['bboxes_ignore'], img_shape, sf, flip))
        if self.with_keypoint:
            kps = keypoint_transform(ann['keypoints'], img_shape, ann['labels'], sf, self.flip_pairs, flip)
            if self.group_mode:   # copy a labelled landmark to the unlabelled members of its group (custom.py:279-286)
                for inst in kps:
                    for group in self.keypoint_groups:
                        vis = inst[group, 2] > 0
                        if vis.sum() > 0:
                            inst[group, :] = inst[group, :][np.tile(vis[:, None], (1, 3))]
            data['gt_keypoints'] = torch.from_numpy(kps.astype(np.float32))
        return data

    def prepare_test_img(self, idx, img=None):
        info = self.img_infos[idx]
        img = self.load_image(idx) if img is None else img
        imgs, metas = [], []
        for scale in self.img_scales:
            for flip in ([False, True] if self.flip_ratio > 0 else [False]):
                t, img_shape, pad_shape, sf = self.img_transform(img, scale, flip, keep_ratio=self.resize_keep_ratio)
                imgs.append(torch.from_numpy(t))
                metas.append(self._meta(info, img_shape, pad_shape, sf, flip))
        return dict(img=imgs, img_meta=metas)

    def __getitem__(self, idx):
        if self.test_mode:
            return self.prepare_test_img(idx)
        while True:
            data = self.prepare_train_img(idx)
            if data is not None:
                return data
            idx = int(np.random.choice(np.where(self.flag == self.flag[idx])[0]))


def collate(batch):
    """Training samples -> one batch: images zero-padded (bottom/right) to the largest in the batch and stacked,
    everything else as per-image lists (what ``RepPointsDetectorKp.forward_train`` takes)."""
    H = max(b['img'].shape[1] for b in batch)
    W = max(b['img'].shape[2] for b in batch)
    img = batch[0]['img'].new_zeros((len(batch), 3, H, W))
    for i, b in enumerate(batch):
        img[i, :, :b['img'].shape[1], :b['img'].shape[2]] = b['img']
    out = dict(img=img, img_meta=[b['img_meta'] for b in batch])
    for key in batch[0]:
        if key not in ('img', 'img_meta'):
            out[key] = [b[key] for b in batch]
    return out


class GroupSampler(object):
    """Batches of ``samples_per_gpu`` indices from ONE aspect-ratio group (numpy global RNG, like the reference)."""

    def __init__(self, dataset, samples_per_gpu=1):
        self.flag = dataset.flag.astype(np.int64)
        self.samples_per_gpu = samples_per_gpu
        self.group_sizes = np.bincount(self.flag)
        self.num_samples = sum(int(np.ceil(s / samples_per_gpu)) * samples_per_gpu for s in self.group_sizes)

    def __iter__(self):
        spg, runs = self.samples_per_gpu, []
        for g, size in enumerate(self.group_sizes):
            if size == 0:
                continue
            idx = np.where(self.flag == g)[0]
            np.random.shuffle(idx)
            extra = int(np.ceil(size / spg)) * spg - len(idx)
            runs.append(np.concatenate([idx, idx[:extra]]))
        flat = np.concatenate(runs)
        order = np.random.permutation(range(len(flat) // spg))
        return iter(np.concatenate([flat[i * spg:(i + 1) * spg] for i in order]).astype(np.int64).tolist())

    def __len__(self):
        return self.num_samples


class DistributedGroupSampler(object):
    """Per-rank slice of the epoch's group-homogeneous batches; shuffling seeded by the epoch on every rank."""

    def __init__(self, dataset, samples_per_gpu=1, num_replicas=None, rank=None):
        if num_replicas is None or rank is None:
            import torch.distributed as dist
            on = dist.is_available() and dist.is_initialized()
            num_replicas = (dist.get_world_size() if on else 1) if num_replicas is None else num_replicas
            rank = (dist.get_rank() if on else 0) if rank is None else rank
        self.flag = dataset.flag
        self.samples_per_gpu, self.num_replicas, self.rank, self.epoch = samples_per_gpu, num_replicas, rank, 0
        self.group_sizes = np.bincount(self.flag)
        per = samples_per_gpu * num_replicas
        self.num_samples = sum(int(math.ceil(s / per)) * samples_per_gpu for s in self.group_sizes)
        self.total_size = self.num_samples * num_replicas

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.epoch)
        spg, per = self.samples_per_gpu, self.samples_per_gpu * self.num_replicas
        indices = []
        for grp, size in enumerate(self.group_sizes):
            if size == 0:
                continue
            idx = np.where(self.flag == grp)[0]
            idx = idx[torch.randperm(int(size), generator=g).numpy()].tolist()
            idx += idx[:int(math.ceil(size / per)) * per - len(idx)]
            indices += idx
        assert len(indices) == self.total_size
        order = torch.randperm(len(indices) // spg, generator=g).tolist()
        indices = [indices[j] for i in order for j in range(i * spg, (i + 1) * spg)]
        lo = self.num_samples * self.rank
        return iter(indices[lo:lo + self.num_samples])

    def __len__(self):
        return self.num_samples


def build_dataset(cfg, default_args=None):
    from .registry import build_from_cfg
    return build_from_cfg(cfg, DATASETS, default_args)
