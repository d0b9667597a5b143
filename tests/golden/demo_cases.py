"""BASELINE config 1 (kgdet_moment_r50_fpn_1x-demo.py on the 32-image demo set) as a reproducible test case.

The reference's JPEGs do not travel; every demo image is RENDERED from its annotation record: image size from
tests/golden/demo_dataset-32.json (reference-held annotation data), a seeded low-frequency background, one textured
patch per annotated garment, seeded fine noise.  Weights are the seeded random initialisation of this repo's
detector (no network for modelzoo://resnet50) with the same score spreading as tests/golden/ref_cases.kgdet_head.
Shared by tests/golden/make_demo_golden.py (REFERENCE detector modules, build container) and the tests."""
import os

import numpy as np
import torch

from kgdet_amd import configs, datasets
from tests.golden import ref_cases

ANN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'demo_dataset-32.json')
IMG_NORM = dict(mean=[154.992, 146.197, 140.744], std=[62.757, 64.507, 62.076], to_rgb=True)
IMG_SCALE = (1333, 800)


def render_image(info, anns, seed):
    rng = np.random.default_rng(seed)
    h, w = int(info['height']), int(info['width'])
    coarse = rng.integers(40, 216, (h // 32 + 2, w // 32 + 2, 3)).astype(np.float32)
    img = np.kron(coarse, np.ones((32, 32, 1), np.float32))[:h, :w]
    for a in anns:
        x, y, bw, bh = [int(round(v)) for v in a['bbox']]
        x0, y0, x1, y1 = max(x, 0), max(y, 0), min(x + bw, w), min(y + bh, h)
        if x1 <= x0 or y1 <= y0:
            continue
        colour = np.array([(37 * a['category_id']) % 256, (91 * a['category_id'] + 60) % 256,
                           (173 * a['category_id'] + 120) % 256], np.float32)
        yy, xx = np.mgrid[y0:y1, x0:x1]
        stripes = 25.0 * np.sin((xx + 2 * yy) * (0.05 + 0.01 * a['category_id']))[..., None]
        img[y0:y1, x0:x1] = 0.35 * img[y0:y1, x0:x1] + 0.65 * colour + stripes
    img += rng.normal(0, 6.0, img.shape).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def demo_dataset(test_mode=True, **kw):
    """DeepFashion2Dataset with the demo config's test (or train) pipeline settings and rendered pixels"""
    args = dict(ann_file=ANN, img_prefix='/rendered/', img_scale=IMG_SCALE, img_norm_cfg=IMG_NORM, size_divisor=32,
                flip_ratio=0, with_mask=False, with_crowd=False, with_label=not test_mode, with_keypoint=True,
                test_mode=test_mode)
    args.update(kw)
    data = datasets.DeepFashion2Dataset(**args)

    def load_image(idx):
        info = data.img_infos[idx]
        anns = data.coco.load_anns(data.coco.get_ann_ids(img_ids=[info['id']]))
        return render_image(info, anns, seed=1000 + idx)
    data.load_image = load_image
    return data


def spread_head(head):
    """the deterministic score / point spreading of ref_cases.kgdet_head applied to a detector's head"""
    g = torch.Generator().manual_seed(1)
    for blk in (head.kp_rep_block_1, head.kp_rep_block_2, head.kp_rep_block_3):
        blk.reppts_out.weight.data.normal_(0, 0.03, generator=g)
        blk.keypts_out.weight.data.normal_(0, 0.03, generator=g)
        blk.cls_out.weight.data.normal_(0, 0.05, generator=g)
    head.kp_rep_block_3.cls_out.bias.data.fill_(-4.3)
    head.moment_transfer.data = torch.tensor([0.2, -0.1])


def demo_detector():
    from kgdet_amd.registry import build_detector
    cfg = configs.kgdet_r50_fpn()
    torch.manual_seed(0)
    model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    spread_head(model.bbox_head)
    return cfg, model


def category_slices(data):
    return data.gt_class_keypoints_dict       # label (1-based) -> (lo, hi) landmark range
