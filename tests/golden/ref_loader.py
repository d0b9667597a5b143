"""Fixture-generation harness (build container only): import the reference's pure-PyTorch host logic
from where it lies under /root/reference, WITHOUT copying it and without running `mmdet/__init__`.

What executes is the reference's own code, file by file:
    mmdet/core/anchor/point_generator.py, point_target_kp.py
    mmdet/core/bbox/geometry.py, assign_sampling.py, assigners/*, samplers/*
    mmdet/core/utils/misc.py (multi_apply), mmdet/core/post_processing/bbox_nms_kp.py
    mmdet/ops/nms/nms_wrapper.py over the COMPILED reference nms_cpu.cpp / soft_nms_cpu.pyx (built by oracle/build_ref.py into $KGDET_REF_BUILD, outside the tree)
    mmdet/utils/registry.py, mmdet/models/{registry,builder}.py, models/utils/*, models/losses/*
    mmdet/models/anchor_heads/reppoints_head_kp3rep_cas_1_assign_once.py, reppoints_head_kp_serial.py, ...
The package objects `mmdet`, `mmdet.core`, ... are empty modules whose `__path__` points at the real
directories (the trick oracle/build_ref.py uses for pycocotools), so relative imports resolve to the real files.

What is NOT the reference and is supplied here as glue, because the image lacks it:
  * `mmcv` (pinned by the reference at mmcv>=0.2.10, setup.py:146): only the non-algorithmic helpers the files
    above touch -- `is_str`, `cnn.{normal,constant,kaiming,xavier,uniform}_init`, `runner.obj_from_dict`,
    `runner.load_checkpoint` (unused) -- restated from mmcv 0.2.x's published behaviour.  No fixture value
    depends on the init helpers: every weight is overwritten from a seeded state_dict.
  * `mmdet.ops.DeformConv` / `ModulatedDeformConv` / `sigmoid_focal_loss`: the reference implements these only
    in CUDA.  They are bound to the test-side CPU formulations (tests/torch_ref.py: grid_sample + einsum
    deformable conv, and the focal formula of sigmoid_focal_loss_cuda.cu) -- so fixtures made through them pin
    the HEAD / TARGET / LOSS / DECODE / NMS logic to the reference, not the deformable-conv arithmetic
    (that stays "parity unpinned").
Nothing here travels to the GPU box or is imported by the product; tests read only the .npz files.
"""
import importlib
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
REF = '/root/reference/mmdetection/mmdet'


def available():
    return os.path.isfile(os.path.join(REF, 'core', 'anchor', 'point_target_kp.py'))


def _bare(name, path=None):
    m = types.ModuleType(name)
    if path is not None:
        m.__path__ = [path]
    sys.modules[name] = m
    parent, _, child = name.rpartition('.')
    if parent:
        setattr(sys.modules[parent], child, m)
    return m


def _mmcv_glue():
    try:
        import mmcv  # noqa: F401  (a real mmcv wins if one is ever installed)
        return
    except ImportError:
        pass
    mmcv = _bare('mmcv')
    cnn = _bare('mmcv.cnn')
    runner = _bare('mmcv.runner')
    mmcv.is_str = lambda x: isinstance(x, str)

    def dump(obj, file, **kw):                   # mmcv.dump for the '.json' result files of coco_utils.results2json
        import json
        assert file.endswith('.json')
        with open(file, 'w') as f:
            json.dump(obj, f)
    mmcv.dump = dump

    def constant_init(module, val, bias=0):
        nn.init.constant_(module.weight, val)
        if getattr(module, 'bias', None) is not None:
            nn.init.constant_(module.bias, bias)

    def normal_init(module, mean=0, std=1, bias=0):
        nn.init.normal_(module.weight, mean, std)
        if getattr(module, 'bias', None) is not None:
            nn.init.constant_(module.bias, bias)

    def xavier_init(module, gain=1, bias=0, distribution='normal'):
        (nn.init.xavier_uniform_ if distribution == 'uniform' else nn.init.xavier_normal_)(module.weight, gain=gain)
        if getattr(module, 'bias', None) is not None:
            nn.init.constant_(module.bias, bias)

    def uniform_init(module, a=0, b=1, bias=0):
        nn.init.uniform_(module.weight, a, b)
        if getattr(module, 'bias', None) is not None:
            nn.init.constant_(module.bias, bias)

    def kaiming_init(module, mode='fan_out', nonlinearity='relu', bias=0, distribution='normal'):
        f = nn.init.kaiming_uniform_ if distribution == 'uniform' else nn.init.kaiming_normal_
        f(module.weight, mode=mode, nonlinearity=nonlinearity)
        if getattr(module, 'bias', None) is not None:
            nn.init.constant_(module.bias, bias)

    def obj_from_dict(info, parent=None, default_args=None):
        args = dict(info)
        obj_type = args.pop('type')
        if isinstance(obj_type, str):
            obj_type = getattr(parent, obj_type) if parent is not None else sys.modules[obj_type]
        for k, v in (default_args or {}).items():
            args.setdefault(k, v)
        return obj_type(**args)

    def load_checkpoint(*a, **k):
        raise RuntimeError('fixtures never load checkpoints')

    cnn.constant_init, cnn.normal_init, cnn.xavier_init = constant_init, normal_init, xavier_init
    cnn.uniform_init, cnn.kaiming_init = uniform_init, kaiming_init
    runner.obj_from_dict, runner.load_checkpoint = obj_from_dict, load_checkpoint


class _CpuDeformConv(nn.Module):
    """constructor / parameter contract of mmdet/ops/dcn/deform_conv.py:190-236, CPU arithmetic from tests/torch_ref"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False):
        super().__init__()
        assert not bias and groups == 1
        k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        self.stride, self.padding, self.dilation, self.deformable_groups = stride, padding, dilation, deformable_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, k, k))
        nn.init.normal_(self.weight, 0, 0.01)

    def forward(self, x, offset):
        from tests import torch_ref
        # .contiguous(): the CUDA op returns a dense NCHW tensor (deform_conv.py:40-41) and the reference relies on it
        # (`.view` in points2kpt); the einsum formulation may hand back permuted strides
        return torch_ref.deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation).contiguous()


_LOADED = {}


def load():
    """-> namespace of reference objects (classes / functions executed from /root/reference in place)."""
    if _LOADED:
        return _LOADED['ns']
    assert available(), 'the reference checkout is not mounted: fixtures can only be generated in the build container'
    from tests import torch_ref
    _mmcv_glue()
    _bare('mmdet', REF)
    for sub in ('core', 'core/anchor', 'core/bbox', 'core/utils', 'core/post_processing', 'models',
                'models/anchor_heads', 'ops', 'ops/nms', 'utils'):
        _bare('mmdet.' + sub.replace('/', '.'), os.path.join(REF, sub))
    imp = importlib.import_module

    # ops the reference has only in CUDA -> CPU formulations (see the module docstring)
    ops = sys.modules['mmdet.ops']
    ops.DeformConv = _CpuDeformConv
    ops.sigmoid_focal_loss = lambda pred, target, gamma, alpha: torch_ref.py_sigmoid_focal_loss(pred, target, gamma, alpha)
    # NMS: the reference's own wrapper over the reference's own compiled CPU kernels
    from oracle import build_ref
    ref_nms_cpu, ref_soft_nms = build_ref.load()
    sys.modules['mmdet.ops.nms.nms_cpu'] = ref_nms_cpu
    sys.modules['mmdet.ops.nms'].nms_cpu = ref_nms_cpu
    soft_mod = types.ModuleType('mmdet.ops.nms.soft_nms_cpu')
    soft_mod.soft_nms_cpu = ref_soft_nms
    sys.modules['mmdet.ops.nms.soft_nms_cpu'] = soft_mod
    _bare('mmdet.ops.nms.nms_cuda')          # never reached with CPU tensors
    nms_wrapper = imp('mmdet.ops.nms.nms_wrapper')
    ops.nms, ops.soft_nms = nms_wrapper.nms, nms_wrapper.soft_nms

    core = sys.modules['mmdet.core']
    geometry = imp('mmdet.core.bbox.geometry')
    assigners = imp('mmdet.core.bbox.assigners')            # real __init__: pure torch
    samplers = imp('mmdet.core.bbox.samplers')              # real __init__
    assign_sampling = imp('mmdet.core.bbox.assign_sampling')
    bbox = sys.modules['mmdet.core.bbox']
    for src in (assigners, samplers, assign_sampling, geometry):
        for k in dir(src):
            if not k.startswith('_'):
                setattr(bbox, k, getattr(src, k))
    misc = imp('mmdet.core.utils.misc')
    sys.modules['mmdet.core.utils'].multi_apply = misc.multi_apply
    sys.modules['mmdet.core.utils'].unmap = misc.unmap
    point_generator = imp('mmdet.core.anchor.point_generator')
    # the reference's grid_points / valid_flags default to device='cuda' and the heads rely on the default:
    # here everything runs on the CPU, so only the DEFAULT ARGUMENT is changed (no code)
    PG = point_generator.PointGenerator
    PG.grid_points.__defaults__ = tuple('cpu' if d == 'cuda' else d for d in PG.grid_points.__defaults__)
    PG.valid_flags.__defaults__ = tuple('cpu' if d == 'cuda' else d for d in PG.valid_flags.__defaults__)
    point_target_kp = imp('mmdet.core.anchor.point_target_kp')
    point_target = imp('mmdet.core.anchor.point_target')
    bbox_nms_kp = imp('mmdet.core.post_processing.bbox_nms_kp')
    core.PointGenerator = point_generator.PointGenerator
    core.multi_apply = misc.multi_apply
    core.multiclass_nms_kp = bbox_nms_kp.multiclass_nms_kp
    core.point_target_kp = point_target_kp.point_target_kp
    core.point_target = point_target.point_target
    core.bbox_overlaps = geometry.bbox_overlaps

    imp('mmdet.utils.registry')
    utils = sys.modules['mmdet.utils']
    utils.Registry = sys.modules['mmdet.utils.registry'].Registry
    utils.build_from_cfg = sys.modules['mmdet.utils.registry'].build_from_cfg
    imp('mmdet.models.registry')
    builder = imp('mmdet.models.builder')
    imp('mmdet.models.utils')                                # real __init__
    losses = imp('mmdet.models.losses')                      # real __init__
    models = sys.modules['mmdet.models']
    models.builder, models.losses = builder, losses
    head_kgdet = imp('mmdet.models.anchor_heads.reppoints_head_kp3rep_cas_1_assign_once')
    head_serial = imp('mmdet.models.anchor_heads.reppoints_head_kp_serial')
    head_parallel = imp('mmdet.models.anchor_heads.reppoints_head_kp_parallel')

    ns = types.SimpleNamespace(
        PointGenerator=point_generator.PointGenerator, point_target_kp=point_target_kp.point_target_kp,
        point_target=point_target.point_target,
        PointAssigner=assigners.PointAssigner, MaxIoUAssigner=assigners.MaxIoUAssigner,
        bbox_overlaps=geometry.bbox_overlaps, multiclass_nms_kp=bbox_nms_kp.multiclass_nms_kp,
        py_sigmoid_focal_loss=sys.modules['mmdet.models.losses.focal_loss'].py_sigmoid_focal_loss,
        FocalLoss=losses.FocalLoss, SmoothL1Loss=losses.SmoothL1Loss, smooth_l1_loss=losses.smooth_l1_loss,
        build_head=builder.build_head, build_loss=builder.build_loss,
        head_kgdet=head_kgdet, head_serial=head_serial, head_parallel=head_parallel,
        nms=nms_wrapper.nms, soft_nms=nms_wrapper.soft_nms)
    _LOADED['ns'] = ns
    return ns


def load_detector():
    """load() + the reference's detector stack, executed in place: models/backbones/resnet.py, models/necks/fpn2.py
    (+ fpn.py), models/detectors/{base,single_stage,reppoints_detector_kp}.py, core/fp16/decorators.py,
    core/bbox/transforms.py, core/post_processing/bbox_nms.py, core/evaluation/class_names.py.
    Extra glue: `mmdet.ops.ContextBlock` / `ModulatedDeformConv` are placeholders that are never instantiated by
    the KGDet configs; `pycocotools.mask` is the reference's own (oracle/build_ref.py build, outside the tree)."""
    ns = load()
    if hasattr(ns, 'build_detector'):
        return ns
    imp = importlib.import_module
    from oracle import build_ref
    build_ref.load_reference_evaluator()                    # registers the reference's pycocotools (+ compiled _mask)
    if 'terminaltables' not in sys.modules:                 # table printing of recall.py / mean_ap.py: never reached
        try:
            import terminaltables  # noqa: F401
        except ImportError:
            _bare('terminaltables').AsciiTable = None
    ops, core = sys.modules['mmdet.ops'], sys.modules['mmdet.core']

    class _NotInKGDet(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError('not on the KGDet path')
    ops.ContextBlock = ops.ModulatedDeformConv = _NotInKGDet
    for sub in ('core/fp16', 'core/evaluation', 'models/backbones', 'models/necks', 'models/detectors',
                'models/plugins'):
        _bare('mmdet.' + sub.replace('/', '.'), os.path.join(REF, sub))
    dec = imp('mmdet.core.fp16.decorators')
    core.auto_fp16, core.force_fp32 = dec.auto_fp16, dec.force_fp32
    transforms = imp('mmdet.core.bbox.transforms')
    for k in ('bbox2result', 'bbox_mapping_back', 'bbox_mapping', 'bbox2roi', 'bbox_flip'):
        setattr(core, k, getattr(transforms, k))
    core.multiclass_nms = imp('mmdet.core.post_processing.bbox_nms').multiclass_nms
    core.get_classes = imp('mmdet.core.evaluation.class_names').get_classes
    core.tensor2imgs = sys.modules['mmdet.core.utils.misc'].tensor2imgs
    ga = imp('mmdet.models.plugins.generalized_attention')
    sys.modules['mmdet.models.plugins'].GeneralizedAttention = ga.GeneralizedAttention
    resnet = imp('mmdet.models.backbones.resnet')
    fpn2 = imp('mmdet.models.necks.fpn2')
    fpn = imp('mmdet.models.necks.fpn')
    imp('mmdet.models.detectors.base')
    imp('mmdet.models.detectors.single_stage')
    det = imp('mmdet.models.detectors.reppoints_detector_kp')
    ns.ResNet, ns.FPN2, ns.FPN, ns.RepPointsDetectorKp = resnet.ResNet, fpn2.FPN2, fpn.FPN, det.RepPointsDetectorKp
    ns.coco_utils = imp('mmdet.core.evaluation.coco_utils')     # results2json / coco_eval of tools/test.py
    ns.build_detector = sys.modules['mmdet.models.builder'].build_detector
    return ns


if __name__ == '__main__':
    ns = load_detector()
    print('reference host logic loaded in place:', sorted(k for k in vars(ns)))
