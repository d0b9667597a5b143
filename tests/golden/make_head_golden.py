"""Generate tests/golden/head_golden.npz: a seeded small KGDet head (32 channels, 8x10 map, B=2)
run through the build's CPU path (tests/cpu_ops.py = grid_sample/einsum deformable conv, torch moment
bbox, the reference's pure-torch focal formula) in float64.  Stored (inputs and weights are re-created from seeds): the nine forward maps (keypoint maps subsampled), the nine loss values and the decoded pre-NMS boxes/scores.
The GPU test loads the same weights into the HIP-backed head and must reproduce them (1e-3 rel on
coordinates is the north-star bar; the test uses 2e-4).

    python tests/golden/make_head_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from kgdet_amd import configs, synthetic  # noqa: E402
from kgdet_amd.registry import build_head  # noqa: E402
from tests import cpu_ops  # noqa: E402


KPT_STRIDE = 7


def small_head():
    torch.manual_seed(0)
    cfg = configs.kgdet_r50_fpn().model.bbox_head.copy()
    cfg.update(in_channels=32, feat_channels=32, point_feat_channels=32,
               norm_cfg=dict(type='GN', num_groups=8, requires_grad=True))
    head = build_head(cfg)
    head.init_weights()
    # the default N(0, 0.01) init keeps every reppoint within 0.05 px of its centre; spread them so the
    # deformable taps actually move (and some leave the 8x10 map)
    g = torch.Generator().manual_seed(1)
    for blk in (head.kp_rep_block_1, head.kp_rep_block_2, head.kp_rep_block_3):
        blk.reppts_out.weight.data.normal_(0, 0.08, generator=g)
        blk.keypts_out.weight.data.normal_(0, 0.08, generator=g)
    head.moment_transfer.data = torch.tensor([0.2, -0.1])
    return head


def make_inputs():
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 32, 8, 10, generator=g)
    batch = synthetic.make_batch(2, 'cpu', seed=5, img_shape=(256, 320, 3), pad_shape=(256, 320, 3))
    for k in ('gt_bboxes', 'gt_keypoints'):
        batch[k] = [t.clamp(max=250) for t in batch[k]]
    return x, batch


def main():
    head = small_head().double()
    x, batch = make_inputs()
    out = {}  # inputs and weights are re-created from their seeds by the test (small_head / make_inputs)
    names = ['cls_1', 'cls_2', 'cls_3', 'kpt_1', 'kpt_2', 'kpt_3', 'bbox_1', 'bbox_2', 'bbox_3']
    cfg = configs.kgdet_r50_fpn()
    with cpu_ops.patched():
        # float64 forward; point grid etc. are float32 in the head, so cast GT to double for the loss
        outs = head([x.double()], batch['img_meta'])
        for n, o in zip(names, outs):
            a = o[0].detach().numpy().astype(np.float32)
            out['out:' + n] = a[:, ::KPT_STRIDE] if n.startswith('kpt') else a   # every 7th keypoint channel
        head_f = small_head()
        outs_f = head_f([x], batch['img_meta'])
        losses = head_f.loss(*outs_f, batch['gt_bboxes'], batch['gt_labels'], batch['gt_keypoints'],
                             batch['img_meta'], cfg.train_cfg)
        for k, v in losses.items():
            out['loss:' + k] = np.float64(sum(float(t) for t in v))
        head_f.eval()
        with torch.no_grad():
            res = head_f.get_bboxes(*head_f([x], batch['img_meta']), batch['img_meta'], cfg.test_cfg, rescale=True,
                                    nms=False)
        out['dec:bboxes'] = np.stack([r[0].numpy() for r in res])
        out['dec:scores'] = np.stack([r[1].numpy() for r in res])
        out['dec:kpts'] = np.stack([r[2].numpy() for r in res])[:, :, ::KPT_STRIDE * 3]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'head_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes;',
          {k: float(v) for k, v in out.items() if k.startswith('loss:')})


if __name__ == '__main__':
    main()
