"""kgdet_amd -- MI355X-native implementation of KGDet's data-parallel hot path.

Host side mirrors the reference's mmdetection operator / registry API; the compute is in
libkgdet_hip.so (hand-written HIP for gfx950, C ABI in include/kgdet_hip.h).
"""
__version__ = '0.1.0'
