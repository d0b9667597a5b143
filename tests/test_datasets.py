"""kgdet_amd.datasets (data side): closed-form cases for the image / box / landmark transforms
(mmdet/datasets/transforms.py), the demo annotation fixture through DeepFashion2Dataset, collate and the samplers
(mmdet/datasets/loader/sampler.py).  The reference pipeline needs mmcv + cv2 (absent) -- parity unpinned by
reference outputs, see the module docstring."""
import os

import numpy as np
import pytest
import torch

from kgdet_amd import datasets as ds
from kgdet_amd.evaluation import landmark_meta

HERE = os.path.dirname(os.path.abspath(__file__))
ANN = os.path.join(HERE, 'golden', 'demo_dataset-32.json')
NORM = dict(mean=[154.992, 146.197, 140.744], std=[62.757, 64.507, 62.076], to_rgb=True)


def _dataset(**kw):
    args = dict(ann_file=ANN, img_prefix='/nonexistent/', img_scale=(1333, 800), img_norm_cfg=NORM, size_divisor=32,
                flip_ratio=0, with_keypoint=True, with_mask=False, with_crowd=False, with_label=True)
    args.update(kw)
    return ds.DeepFashion2Dataset(**args)


def _fake_image(info, seed=0):
    return np.random.default_rng(seed).integers(0, 256, (info['height'], info['width'], 3), dtype=np.uint8)


def test_rescale_rule_matches_mmcv_imrescale():
    assert ds.rescale_size(750, 500, (1333, 800))[:2] == (1200, 800)        # short edge binds
    assert ds.rescale_size(400, 1000, (1333, 800))[:2] == (533, 1333)       # long edge binds: int(400*1.333+.5)
    h, w, f = ds.rescale_size(1024, 683, (1333, 800))
    assert (h, w) == (int(1024 * f + 0.5), int(683 * f + 0.5)) and abs(f - 800 / 683) < 1e-12


def test_bilinear_resize_geometry():
    # a linear ramp is reproduced exactly by bilinear interpolation away from the clamped border
    ramp = np.tile(np.arange(0, 200, 2, dtype=np.uint8)[None, :, None], (10, 1, 3))        # value = 2 * x
    out = ds.resize_bilinear_u8(ramp, 10, 200)                                             # x' -> x = (x'+.5)/2 - .5
    want = np.clip(np.round(2 * ((np.arange(200) + 0.5) / 2 - 0.5)), 0, 198)
    assert np.array_equal(out[3, 2:-2, 0], want[2:-2].astype(np.uint8))
    # 2x2 -> 4x4: corners keep their value, the centre 2x2 are the 9:3:3:1 blends
    src = np.array([[0, 100], [200, 40]], dtype=np.uint8)[:, :, None].repeat(3, 2)
    up = ds.resize_bilinear_u8(src, 4, 4)[:, :, 0].astype(int)
    assert up[0, 0] == 0 and up[0, 3] == 100 and up[3, 0] == 200 and up[3, 3] == 40
    assert up[1, 1] == round((9 * 0 + 3 * 100 + 3 * 200 + 1 * 40) / 16)


def test_image_transform_normalise_flip_pad():
    t = ds.ImageTransform(size_divisor=32, **NORM)
    img = np.zeros((100, 150, 3), np.uint8)
    img[..., 0], img[..., 1], img[..., 2] = 10, 20, 30
    img[:, :75, 0] = 200                                                     # left half red-ish
    out, img_shape, pad_shape, sf = t(img, (300, 200), flip=False)
    assert img_shape == (200, 300, 3) and pad_shape == (224, 320, 3) and out.shape == (3, 224, 320) and sf == 2.0
    assert np.allclose(out[1, :200, :300], (20 - NORM['mean'][1]) / NORM['std'][1], atol=1e-6)
    assert np.all(out[:, 200:, :] == 0) and np.all(out[:, :, 300:] == 0)     # zero padding AFTER normalisation
    flipped = t(img, (300, 200), flip=True)[0]
    assert np.allclose(flipped[:, :200, :300], out[:, :200, :300][:, :, ::-1])
    assert flipped[0, 50, 299] > flipped[0, 50, 0]                           # the red half moved to the right


def test_bbox_and_keypoint_flip_are_involutions_and_swap_partners():
    meta = landmark_meta()
    shape = (800, 1200, 3)
    boxes = np.array([[10, 20, 300, 400], [500, 100, 1500, 900]], np.float32)
    once = ds.bbox_transform(boxes, shape, 1.0, flip=True)
    assert np.array_equal(once[0], [1200 - 300 - 1, 20, 1200 - 10 - 1, 400])
    assert once[1].tolist() == [0, 100, 699, 799]                            # clipped to img_shape - 1
    twice = ds.bbox_transform(once[:1], shape, 1.0, flip=True)
    assert np.array_equal(twice[0], boxes[0])
    label = 2                                                                # long_sleeved_shirt: landmarks 25..57
    lo, hi = meta['landmark_ranges'][label - 1]
    kp = np.zeros((294, 3))
    kp[lo:hi, 0] = np.arange(hi - lo) * 10 + 5
    kp[lo:hi, 1] = 7
    kp[lo:hi, 2] = 2
    f1 = ds.keypoint_transform([kp], shape, [label], 1.0, meta['swap_pairs'], flip=True)[0]
    a, b = meta['swap_pairs'][label - 1][0]
    assert f1[a, 0] == 1200 - kp[b, 0] - 1 and f1[b, 0] == 1200 - kp[a, 0] - 1 and f1[a, 2] == 2
    f2 = ds.keypoint_transform([f1], shape, [label], 1.0, meta['swap_pairs'], flip=True)[0]
    assert np.array_equal(f2[lo:hi], kp[lo:hi])
    scaled = ds.keypoint_transform([kp], shape, [label], 1.5, meta['swap_pairs'])[0]
    assert np.array_equal(scaled[:, :2], kp[:, :2] * 1.5) and np.array_equal(scaled[:, 2], kp[:, 2])


def test_landmark_tables_are_consistent():
    meta = landmark_meta()
    assert len(meta['classes']) == 13 and meta['landmark_ranges'][0] == [0, 25] and meta['landmark_ranges'][-1][1] == 294
    for c, pairs in enumerate(meta['swap_pairs']):
        lo, hi = meta['landmark_ranges'][c]
        flat = [i for p in pairs for i in p]
        assert all(lo <= i < hi for i in flat) and len(set(flat)) == len(flat)     # disjoint pairs inside the range
    d = _dataset()
    perm = d.flip_indices.reshape(294, 2)
    assert np.array_equal(perm[:, 1], perm[:, 0] + 1) and np.array_equal(np.sort(perm[:, 0] // 2), np.arange(294))
    assert np.array_equal(perm[perm[:, 0] // 2, 0] // 2, np.arange(294))            # the permutation is an involution
    members = sorted(i for g in meta['groups'] for i in g)
    assert members == list(range(294))                                              # groups partition the landmarks


def test_demo_annotations_through_the_dataset():
    d = _dataset()
    assert len(d) == 32 and d.cat_ids == list(range(1, 14)) and d.flag.shape == (32,)
    n_inst = 0
    for idx in range(len(d)):
        info = d.img_infos[idx]
        sample = d.prepare_train_img(idx, img=_fake_image(info, idx))
        meta = sample['img_meta']
        H, W = meta['img_shape'][:2]
        assert sample['img'].shape[1] % 32 == 0 and sample['img'].shape[2] % 32 == 0 and sample['img'].dtype == torch.float32
        assert max(H, W) <= 1333 and min(H, W) <= 800 and (max(H, W) == 1333 or min(H, W) == 800)
        assert meta['ori_shape'] == (info['height'], info['width'], 3) and meta['flip'] is False
        g = sample['gt_bboxes']
        assert g.shape[1] == 4 and len(sample['gt_labels']) == len(g) == len(sample['gt_keypoints'])
        assert (g[:, 2] >= g[:, 0]).all() and g[:, 0].min() >= 0 and g[:, 2].max() <= W - 1 and g[:, 3].max() <= H - 1
        for lab, kp in zip(sample['gt_labels'].tolist(), sample['gt_keypoints']):
            lo, hi = d.gt_class_keypoints_dict[lab]
            vis = (kp[:, 2] > 0).nonzero().flatten()
            assert 1 <= lab <= 13 and kp.shape == (294, 3) and (len(vis) == 0 or (vis.min() >= lo and vis.max() < hi))
        n_inst += len(g)
    assert n_inst == 55                                                      # 55 instances in the demo set
    t = _dataset(test_mode=True, with_label=False, flip_ratio=0.5).prepare_test_img(0, img=_fake_image(d.img_infos[0]))
    assert len(t['img']) == 2 and t['img_meta'][1]['flip'] is True and torch.equal(
        t['img'][1][:, :t['img_meta'][0]['img_shape'][0], :t['img_meta'][0]['img_shape'][1]],
        t['img'][0][:, :t['img_meta'][0]['img_shape'][0], :t['img_meta'][0]['img_shape'][1]].flip(-1))


def test_flipped_training_sample_mirrors_targets():
    d = _dataset(flip_ratio=1.0)
    plain = _dataset(flip_ratio=0.0)
    img = _fake_image(d.img_infos[3], 3)
    a, b = plain.prepare_train_img(3, img=img), d.prepare_train_img(3, img=img)
    W = a['img_meta']['img_shape'][1]
    assert b['img_meta']['flip'] is True
    assert torch.allclose(b['gt_bboxes'][:, 0], W - a['gt_bboxes'][:, 2] - 1) and torch.equal(b['gt_labels'], a['gt_labels'])
    va, vb = a['gt_keypoints'][..., 2] > 0, b['gt_keypoints'][..., 2] > 0
    assert va.sum() == vb.sum() and va.sum() > 0
    assert torch.allclose(torch.sort(b['gt_keypoints'][..., 0][vb])[0], torch.sort(W - a['gt_keypoints'][..., 0][va] - 1)[0])


def test_collate_pads_to_the_largest_image():
    d = _dataset(img_scale=(333, 200))
    samples = [d.prepare_train_img(i, img=_fake_image(d.img_infos[i], i)) for i in (0, 1, 5)]
    batch = ds.collate(samples)
    H, W = max(s['img'].shape[1] for s in samples), max(s['img'].shape[2] for s in samples)
    assert batch['img'].shape == (3, 3, H, W) and len(batch['gt_bboxes']) == 3 and len(batch['img_meta']) == 3
    h0, w0 = samples[0]['img'].shape[1:]
    assert torch.equal(batch['img'][0, :, :h0, :w0], samples[0]['img']) and batch['img'][0, :, h0:].abs().sum() == 0


def test_real_demo_images_decode_when_reference_is_present():
    root = '/root/reference/data/demo_dataset/image/'
    if not os.path.isdir(root):
        pytest.skip('reference checkout not present')
    d = _dataset(img_prefix=root)
    s = d[0]
    info = d.img_infos[0]
    assert s['img_meta']['ori_shape'] == (info['height'], info['width'], 3)
    assert abs(float(s['img'][:, :s['img_meta']['img_shape'][0], :s['img_meta']['img_shape'][1]].mean())) < 3.0


class _Flags(object):
    def __init__(self, flag):
        self.flag = np.asarray(flag, dtype=np.uint8)


def test_group_sampler_batches_are_homogeneous():
    np.random.seed(0)
    data = _Flags([0] * 7 + [1] * 10)
    s = ds.GroupSampler(data, samples_per_gpu=4)
    idx = list(s)
    assert len(idx) == len(s) == 8 + 12
    for i in range(0, len(idx), 4):
        assert len(set(data.flag[idx[i:i + 4]].tolist())) == 1
    assert set(idx) == set(range(17))


def test_distributed_group_sampler_partitions_an_epoch():
    data = _Flags([0] * 9 + [1] * 14)
    parts = []
    for rank in range(2):
        s = ds.DistributedGroupSampler(data, samples_per_gpu=2, num_replicas=2, rank=rank)
        s.set_epoch(3)
        parts.append(list(s))
        assert len(parts[-1]) == len(s) == (12 + 16) // 2
        for i in range(0, len(parts[-1]), 2):
            assert data.flag[parts[-1][i]] == data.flag[parts[-1][i + 1]]
    assert set(parts[0]) | set(parts[1]) == set(range(23))
    again = ds.DistributedGroupSampler(data, samples_per_gpu=2, num_replicas=2, rank=0)
    again.set_epoch(3)
    assert list(again) == parts[0]
    again.set_epoch(4)
    assert list(again) != parts[0]


def test_dataset_is_registered_and_built_from_the_config_dict():
    from kgdet_amd.registry import DATASETS
    cfg = dict(type='DeepFashion2Dataset', ann_file=ANN, img_prefix='x/', img_scale=(1333, 800), img_norm_cfg=NORM,
               size_divisor=32, flip_ratio=0.5, with_keypoint=True, with_mask=False, with_crowd=False, with_label=True,
               group_mode=False)
    d = ds.build_dataset(cfg)
    assert isinstance(d, DATASETS.get('DeepFashion2Dataset')) and d.flip_ratio == 0.5
    with pytest.raises(NotImplementedError):
        ds.build_dataset(dict(cfg, with_mask=True))
