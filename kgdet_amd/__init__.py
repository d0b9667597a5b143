"""kgdet_amd -- MI355X-native implementation of KGDet's data-parallel hot path.

Host side mirrors the reference's mmdetection operator / registry API; the compute is in
libkgdet_hip.so (hand-written HIP for gfx950, C ABI in include/kgdet_hip.h).
"""
__version__ = '0.1.0'

# importing the submodules populates the registries (HEADS, LOSSES, BACKBONES, NECKS, DETECTORS, DATASETS)
from . import losses, backbone, neck, heads, heads_serial, detector, datasets  # noqa: E402,F401
from .registry import (BACKBONES, DETECTORS, HEADS, LOSSES, NECKS, Config, ConfigDict, Registry,  # noqa: E402,F401
                       build_backbone, build_detector, build_from_cfg, build_head, build_loss, build_neck)
