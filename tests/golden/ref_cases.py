"""Seeded inputs shared by tests/golden/make_ref_golden.py (which feeds them to the REFERENCE's code in the build
container) and by the tests (which feed the same inputs to this repo's code, on the CPU for the host logic and on the
GPU for the HIP path).  Pure data construction: imports nothing from the reference."""
import torch

from kgdet_amd import configs, synthetic

KPT_STRIDE = 7


def _kps_for(boxes, labels, g):
    kps = torch.zeros(boxes.shape[0], 294, 3)
    for i in range(boxes.shape[0]):
        lo, hi = synthetic.CLASS_KEYPOINT_SLICES[int(labels[i])]
        n = hi - lo
        wh = boxes[i, 2:] - boxes[i, :2]
        kps[i, lo:hi, 0] = boxes[i, 0] + torch.rand(n, generator=g) * wh[0]
        kps[i, lo:hi, 1] = boxes[i, 1] + torch.rand(n, generator=g) * wh[1]
        kps[i, lo:hi, 2] = torch.randint(0, 3, (n, ), generator=g).float()     # includes v = 0 (not labelled)
    return kps


def _gts(list_of_boxes, seed):
    g = torch.Generator().manual_seed(seed)
    boxes = [torch.tensor(b, dtype=torch.float32).reshape(-1, 4) for b in list_of_boxes]
    labels = [torch.randint(1, 14, (b.shape[0], ), generator=g) for b in boxes]
    kps = [_kps_for(b, lab, g) for b, lab in zip(boxes, labels)]
    return boxes, labels, kps


def pyramid_featmaps(h, w, strides):
    out = []
    for s in strides:
        out.append((-(-h // s), -(-w // s)))
    return out


def pseudo_boxes(points, img, lvl):
    """box proposals for the MaxIoUAssigner (refine) stage: a deterministic box around every grid point"""
    s = points[:, 2:3]
    half = s * (2.0 + 0.5 * ((torch.arange(points.shape[0], dtype=torch.float32)[:, None] * 0.37 + img + lvl) % 3))
    return torch.cat([points[:, :2] - half, points[:, :2] + half * 1.3], 1)


def target_cases():
    """name -> dict(strides, featmaps, pad_shapes, gt_bboxes, gt_labels, gt_keypoints, cfg[, boxes_as_proposals])"""
    kg = dict(assigner=dict(type='PointAssigner', scale=4, pos_num=25), allowed_border=-1, pos_weight=-1, debug=False)
    cases = {}

    def add(name, strides, featmaps, pad_shapes, boxes, seed, cfg, **kw):
        b, lab, k = _gts(boxes, seed)
        cases[name] = dict(strides=strides, featmaps=featmaps, pad_shapes=pad_shapes, gt_bboxes=b, gt_labels=lab,
                           gt_keypoints=k, cfg=cfg, **kw)

    # KGDet: one level, stride 32, 25x42 (800x1344): one GT per image
    add('kgdet_1gt', [32], [(25, 42)], [(800, 1344, 3)] * 2,
        [[[100., 80., 620., 700.]], [[640., 10., 1300., 420.]]], 11, kg)
    # two overlapping GTs whose 25 nearest points collide (the later / closer GT must win per point)
    add('kgdet_overlap', [32], [(25, 42)], [(800, 1344, 3)] * 2,
        [[[200., 100., 700., 600.], [260., 140., 720., 640.]],
         [[300., 200., 900., 700.], [310., 190., 600., 450.], [900., 400., 1330., 790.]]], 12, kg)
    # the batch is padded beyond this image's pad_shape: some grid points are invalid -> mirrored (masked) path
    add('kgdet_invalid_points', [32], [(25, 42)], [(704, 1120, 3), (800, 1344, 3)],
        [[[50., 60., 500., 640.], [400., 300., 1000., 690.]], [[640., 10., 1300., 420.]]], 13, kg)
    # tiny and huge boxes: the level rule log2(w/scale) clamps to the single level; a degenerate (zero-area) GT
    add('kgdet_extremes', [32], [(25, 42)], [(800, 1344, 3)] * 2,
        [[[10., 10., 14., 13.], [0., 0., 1333., 799.]], [[500., 400., 500., 400.], [620., 380., 680., 470.]]], 14, kg)
    # config 5: five levels, init stage = PointAssigner(pos_num=1), GTs spread over the levels
    strides = [8, 16, 32, 64, 128]
    fm = pyramid_featmaps(256, 320, strides)
    init = dict(assigner=dict(type='PointAssigner', scale=4, pos_num=1), allowed_border=-1, pos_weight=-1, debug=False)
    boxes5 = [[[10., 12., 40., 50.], [60., 30., 200., 180.], [5., 5., 310., 250.], [100., 90., 180., 150.]],
              [[150., 100., 300., 240.], [20., 150., 60., 200.]]]
    add('pyramid_init', strides, fm, [(256, 320, 3)] * 2, boxes5, 15, init)
    # (pos_num > 1 takes the k nearest points: box centres off the grid's symmetry axes, so that no two candidate
    # points are equidistant -- topk's choice among exact ties is unspecified and differs between CPU and GPU)
    boxes5b = [[[10.3, 12.9, 41.1, 50.2], [60.7, 30.1, 201.9, 183.3], [5.2, 5.9, 311.3, 251.7], [100.9, 90.3, 183.1, 152.6]],
               [[150.6, 100.2, 301.3, 243.1], [20.4, 150.9, 63.7, 202.2]]]
    add('pyramid_init_pos3', strides, fm, [(256, 320, 3), (224, 300, 3)], boxes5b, 16,
        dict(init, assigner=dict(type='PointAssigner', scale=4, pos_num=3)))
    refine = dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0,
                                ignore_iof_thr=-1), allowed_border=-1, pos_weight=-1, debug=False)
    add('pyramid_refine', strides, fm, [(256, 320, 3)] * 2, boxes5, 17, refine, boxes_as_proposals=True)
    # an image without ground truth
    add('kgdet_empty_gt', [32], [(25, 42)], [(800, 1344, 3)] * 2, [[[100., 80., 620., 700.]], []], 18, kg)
    return cases


def overlap_boxes():
    g = torch.Generator().manual_seed(21)
    a = torch.rand(50, 4, generator=g) * 200
    a[:, 2:] = a[:, :2] + torch.rand(50, 2, generator=g) * 120 + 1
    b = torch.rand(7, 4, generator=g) * 200
    b[:, 2:] = b[:, :2] + torch.rand(7, 2, generator=g) * 150 + 1
    a[3] = b[2]                       # IoU exactly 1
    a[4] = torch.tensor([1000., 1000., 1010., 1010.])   # overlaps nothing
    return a, b


def overlap_labels():
    return torch.tensor([3, 1, 13, 7, 2, 9, 5])


def max_iou_cases():
    return {'serial_refine': dict(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0, ignore_iof_thr=-1),
            'low_quality': dict(pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3, gt_max_assign_all=True),
            'neg_range': dict(pos_iou_thr=0.6, neg_iou_thr=(0.1, 0.4), min_pos_iou=0.2, gt_max_assign_all=False)}


def point_assigner_cases():
    g = torch.Generator().manual_seed(22)
    ys, xs = torch.meshgrid(torch.arange(12.), torch.arange(15.), indexing='ij')
    lv = []
    for s in (8, 16, 32):
        n = (96 // s) * (128 // s)
        yy, xx = torch.meshgrid(torch.arange(96 // s) * float(s), torch.arange(128 // s) * float(s), indexing='ij')
        lv.append(torch.stack([xx.reshape(-1), yy.reshape(-1), torch.full((n, ), float(s))], 1))
    pts = torch.cat(lv)
    gts = torch.tensor([[4., 4., 30., 40.], [10., 20., 100., 90.], [60., 5., 125., 60.], [20., 20., 52., 52.]])
    labels = torch.randint(1, 14, (4, ), generator=g)
    return {'three_levels_pos1': (pts, gts, labels, dict(scale=4, pos_num=1)),
            'three_levels_pos5': (pts, gts, labels, dict(scale=4, pos_num=5)),
            'scale8_pos3': (pts, gts, None, dict(scale=8, pos_num=3)),
            'no_gt': (pts, gts[:0], None, dict(scale=4, pos_num=3))}


def focal_inputs():
    """2100 points x 13 classes (one image of the KGDet map), labels 0..13 with ~3 % positives, per-point weights"""
    g = torch.Generator().manual_seed(23)
    pred = torch.randn(2100, 13, generator=g) * 3 - 2
    target = torch.zeros(2100, dtype=torch.long)
    pos = torch.randperm(2100, generator=g)[:60]
    target[pos] = torch.randint(1, 14, (60, ), generator=g)
    weight = (torch.rand(2100, generator=g) > 0.1).float()
    return pred, target, weight


def smooth_l1_inputs():
    g = torch.Generator().manual_seed(24)
    pred = torch.randn(300, 4, generator=g)
    target = pred + torch.randn(300, 4, generator=g) * torch.tensor([0.01, 0.1, 0.5, 2.0])
    weight = (torch.rand(300, 1, generator=g) > 0.5).float().expand(-1, 4).contiguous()
    return pred, target, weight


# ---------------------------------------------------------------------------------------------------
def kgdet_head():
    """this repo's KGDet head at full width with seeded weights; its state_dict is what the reference head loads"""
    from kgdet_amd.registry import build_head
    torch.manual_seed(0)
    head = build_head(configs.kgdet_r50_fpn().model.bbox_head.copy())
    head.init_weights()
    g = torch.Generator().manual_seed(1)
    for blk in (head.kp_rep_block_1, head.kp_rep_block_2, head.kp_rep_block_3):
        # N(0, 0.01) keeps every reppoint within a fraction of a pixel of its centre: spread them so that the
        # deformable taps move and some leave the map
        blk.reppts_out.weight.data.normal_(0, 0.03, generator=g)
        blk.keypts_out.weight.data.normal_(0, 0.03, generator=g)
        blk.cls_out.weight.data.normal_(0, 0.05, generator=g)
    head.kp_rep_block_3.cls_out.bias.data.fill_(-3.2)       # so that decode + NMS see candidates above score_thr
    head.moment_transfer.data = torch.tensor([0.2, -0.1])
    return head


def kgdet_inputs():
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 256, 25, 42, generator=g)
    batch = synthetic.make_batch(2, 'cpu', seed=5)
    return x, batch


def flip_metas(img_meta):
    """the batch's img_meta with the dataset's real ``flip_indices`` (588 interleaved keypoint channels: left / right landmark
    swaps per category, kgdet_amd/data/deepfashion2_landmarks.json) instead of the synthetic batch's identity"""
    import numpy as np
    from kgdet_amd.datasets import landmark_meta
    perm = np.arange(294)
    for pairs in landmark_meta()['swap_pairs']:
        for a, b in pairs:
            perm[a], perm[b] = b, a
    idx = np.stack([perm * 2, perm * 2 + 1], axis=1).reshape(-1).tolist()
    assert idx != list(range(588))
    return [dict(m, flip_indices=idx) for m in img_meta]


def serial_head(parallel=False):
    from kgdet_amd.registry import build_head
    torch.manual_seed(0)
    head = build_head(configs.reppoints_kp_r50_fpn(parallel=parallel).model.bbox_head.copy())
    head.init_weights()
    g = torch.Generator().manual_seed(3)
    head.reppts_init_out.weight.data.normal_(0, 0.05, generator=g)
    head.keypts_init_out.weight.data.normal_(0, 0.03, generator=g)
    head.reppts_refine_out.weight.data.normal_(0, 0.03, generator=g)
    head.keypts_refine_out.weight.data.normal_(0, 0.03, generator=g)
    head.cls_refine_out.weight.data.normal_(0, 0.05, generator=g)
    head.cls_refine_out.bias.data.fill_(-3.4)
    head.moment_transfer.data = torch.tensor([0.15, -0.05])
    return head


def serial_inputs(size=(256, 320)):
    """five-level pyramid inputs of the serial / parallel heads; ``size=(384, 512)`` puts 48 x 64 = 3072 pixels on the
    stride-8 level: beyond the 1536-pixel limit of the LDS-plane deformable kernels, i.e. the large-map path"""
    h_img, w_img = size
    g = torch.Generator().manual_seed(4 if size == (256, 320) else 14)
    fm = pyramid_featmaps(h_img, w_img, [8, 16, 32, 64, 128])
    xs = [torch.randn(2, 256, h, w, generator=g) for h, w in fm]
    batch = synthetic.make_batch(2, 'cpu', seed=6, img_shape=(h_img, w_img, 3), pad_shape=(h_img, w_img, 3))
    for k in ('gt_bboxes', 'gt_keypoints'):
        batch[k] = [t.clamp(max=h_img - 6) for t in batch[k]]
    return xs, batch


def soft_nms_test_cfg(test_cfg):
    c = configs.ConfigDict(dict(test_cfg))
    c['nms'] = dict(type='soft_nms', iou_thr=0.5, min_score=0.05)
    return c
