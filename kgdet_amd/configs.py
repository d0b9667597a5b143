"""The model / train / test settings of the reference's KGDet configs as plain dicts, for use on
machines where the reference tree is not mounted (the GPU box).  tests/test_config.py checks them
against R/configs/kgdet_moment_r50_fpn_1x-demo.py and
R/configs/reppoints_moment_serial_r50_fpn_1x-deepfashion2.py whenever the reference is present.
"""
from .registry import ConfigDict


def _focal(w):
    return dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=w)


def _sl1(w):
    return dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=w)


def kgdet_r50_fpn():
    """model, train_cfg, test_cfg of kgdet_moment_r50_fpn_1x-{demo,deepfashion2}.py"""
    norm_cfg = dict(type='GN', num_groups=32, requires_grad=True)
    head = dict(type='RepPointsHeadKp3RepCas1AssignOnce', num_classes=14, in_channels=256, feat_channels=256,
                point_feat_channels=256, stacked_convs=3, num_reppts=25, num_keypts=294, gradient_mul=0.1,
                point_strides=[32], point_base_scale=4, flip_forward=False, norm_cfg=norm_cfg,
                transform_method='moment')
    for stage, w in ((1, 0.5), (2, 0.5), (3, 1.0)):
        head['loss_cls_%d' % stage] = _focal(w)
        head['loss_bbox_%d' % stage] = _sl1(w)
        head['loss_kpt_%d' % stage] = _sl1(w)
    model = dict(
        type='RepPointsDetectorKp', pretrained='modelzoo://resnet50',
        backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                      style='pytorch'),
        neck=dict(type='FPN2', in_channels=[256, 512, 1024, 2048], out_channels=256, start_level=1, end_level=-1,
                  add_extra_convs=True, num_outs=5, select_out=[2], norm_cfg=norm_cfg),
        bbox_head=head)
    train_cfg = dict(uniform=dict(assigner=dict(type='PointAssigner', scale=4, pos_num=25), allowed_border=-1,
                                  pos_weight=-1, debug=False))
    test_cfg = dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05, nms=dict(type='nms', iou_thr=0.5),
                    max_per_img=100)
    return ConfigDict(model=model, train_cfg=train_cfg, test_cfg=test_cfg,
                      optimizer=dict(type='Adam', lr=1e-4),
                      optimizer_config=dict(grad_clip=dict(max_norm=35, norm_type=2)))


def reppoints_kp_r50_fpn(parallel=False, soft_nms=False):
    """model, train_cfg, test_cfg of reppoints_moment_{serial,parallel}_r50_fpn_1x-deepfashion2.py
    (BASELINE config 5; ``soft_nms=True`` swaps the test-time NMS for the soft-NMS stress variant)."""
    norm_cfg = dict(type='GN', num_groups=32, requires_grad=True)
    model = dict(
        type='RepPointsDetectorKp', pretrained='modelzoo://resnet50',
        backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                      style='pytorch'),
        neck=dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, start_level=1,
                  add_extra_convs=True, num_outs=5, norm_cfg=norm_cfg),
        bbox_head=dict(type='RepPointsHeadKpParallel' if parallel else 'RepPointsHeadKpSerial', num_classes=14,
                       in_channels=256, feat_channels=256, point_feat_channels=256, stacked_convs=3, num_reppts=9,
                       num_keypts=294, gradient_mul=0.1, point_strides=[8, 16, 32, 64, 128], point_base_scale=4,
                       norm_cfg=norm_cfg, loss_cls=_focal(1.0),
                       loss_bbox_init=dict(type='SmoothL1Loss', beta=0.11, loss_weight=0.5),
                       loss_bbox_refine=dict(type='SmoothL1Loss', beta=0.11, loss_weight=1.),
                       loss_kpt_init=dict(type='SmoothL1Loss', beta=0.11, loss_weight=2.),
                       loss_kpt_refine=dict(type='SmoothL1Loss', beta=0.11, loss_weight=4.),
                       transform_method='moment'))
    train_cfg = dict(
        init=dict(assigner=dict(type='PointAssigner', scale=4, pos_num=1), allowed_border=-1, pos_weight=-1,
                  debug=False),
        refine=dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0,
                                  ignore_iof_thr=-1), allowed_border=-1, pos_weight=-1, debug=False))
    nms = dict(type='soft_nms', iou_thr=0.5, min_score=0.05) if soft_nms else dict(type='nms', iou_thr=0.5)
    test_cfg = dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05, nms=nms, max_per_img=100)
    return ConfigDict(model=model, train_cfg=train_cfg, test_cfg=test_cfg)
