"""Build the two CPU kernels the reference actually has, from the sources where they lie.

TEST INFRASTRUCTURE ONLY.  Compiles, unmodified,

    /root/reference/mmdetection/mmdet/ops/nms/src/nms_cpu.cpp       -> $KGDET_REF_BUILD/ref_nms_cpu*.so
    /root/reference/mmdetection/mmdet/ops/nms/src/soft_nms_cpu.pyx  -> $KGDET_REF_BUILD/soft_nms_cpu*.so
    /root/reference/deepfashion2_api/PythonAPI/pycocotools/_mask.pyx + common/maskApi.c -> $KGDET_REF_BUILD/_mask*.so
        (the compiled dependency of the reference's pure-Python COCO / COCOeval, which are then imported in place)

with g++ / cython + gcc directly (not the reference's setup.py).  Nothing derived from the reference is
written into the repository: outputs (generated C -- Cython embeds the .pyx text in it --, objects, .so) go to
$KGDET_REF_BUILD (default /tmp/kgdet_ref), OUTSIDE the working tree, so no reference-derived artefact is
snapshotted to the GPU box (round-2 review item 2; tests/test_host_logic.py::test_no_reference_text_in_tree).
The GPU tests use the committed golden vectors only.
The rest of the hot path (deform conv, psroi pooling, focal loss) has no CPU implementation in
the reference (CUDA-only) and is therefore unbuildable here.

Used by tests/golden/make_*_golden.py to generate the committed golden vectors and, where the reference is
mounted, by tests/test_oracle_nms.py / test_evaluation.py for a live cross-check (built on demand).
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get('KGDET_REF_BUILD', '/tmp/kgdet_ref')
REF_NMS = '/root/reference/mmdetection/mmdet/ops/nms/src'


def available():
    return os.path.isfile(os.path.join(REF_NMS, 'nms_cpu.cpp'))


def build_nms_cpu():
    import torch
    from torch.utils import cpp_extension as ce
    os.makedirs(OUT, exist_ok=True)
    ext = sysconfig.get_config_var('EXT_SUFFIX')
    so = os.path.join(OUT, 'ref_nms_cpu' + ext)
    src = os.path.join(REF_NMS, 'nms_cpu.cpp')
    if os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(src):
        return so
    inc = []
    for p in ce.include_paths():
        inc += ['-isystem', p]
    inc += ['-isystem', sysconfig.get_paths()['include']]
    libdir = os.path.join(os.path.dirname(torch.__file__), 'lib')
    cmd = ['g++', '-O2', '-fPIC', '-shared', '-std=c++17', '-w',
           '-DTORCH_EXTENSION_NAME=ref_nms_cpu', '-DTORCH_API_INCLUDE_EXTENSION_H',
           '-D_GLIBCXX_USE_CXX11_ABI=%d' % int(torch._C._GLIBCXX_USE_CXX11_ABI)] + inc + [
        src, '-o', so, '-L' + libdir, '-Wl,-rpath,' + libdir,
        '-lc10', '-ltorch_cpu', '-ltorch', '-ltorch_python']
    subprocess.check_call(cmd)
    return so


def build_soft_nms():
    import numpy as np
    os.makedirs(OUT, exist_ok=True)
    ext = sysconfig.get_config_var('EXT_SUFFIX')
    so = os.path.join(OUT, 'soft_nms_cpu' + ext)
    src = os.path.join(REF_NMS, 'soft_nms_cpu.pyx')
    if os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(src):
        return so
    c_file = os.path.join(OUT, 'soft_nms_cpu.c')
    subprocess.check_call([sys.executable, '-m', 'cython', '-3', src, '-o', c_file])
    subprocess.check_call(['gcc', '-O2', '-fPIC', '-shared', '-w',
                           '-I' + sysconfig.get_paths()['include'], '-I' + np.get_include(),
                           c_file, '-o', so])
    return so


REF_API = '/root/reference/deepfashion2_api'


def build_mask_api():
    """pycocotools' only compiled part (_mask.pyx + common/maskApi.c), built where the sources lie."""
    import numpy as np
    os.makedirs(OUT, exist_ok=True)
    ext = sysconfig.get_config_var('EXT_SUFFIX')
    so = os.path.join(OUT, '_mask' + ext)
    pyx = os.path.join(REF_API, 'PythonAPI', 'pycocotools', '_mask.pyx')
    if os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(pyx):
        return so
    c_file = os.path.join(OUT, '_mask.c')
    subprocess.check_call([sys.executable, '-m', 'cython', '-3', '-I', os.path.join(REF_API, 'common'), pyx,
                           '-o', c_file])
    subprocess.check_call(['gcc', '-O2', '-fPIC', '-shared', '-w', '-std=c99',
                           '-I' + sysconfig.get_paths()['include'], '-I' + np.get_include(),
                           '-I' + os.path.join(REF_API, 'common'),
                           c_file, os.path.join(REF_API, 'common', 'maskApi.c'), '-o', so])
    return so


def load_reference_evaluator():
    """(COCO, COCOeval) classes of the reference's DeepFashion2 evaluator, imported from where they lie
    (deepfashion2_api/PythonAPI/pycocotools/{coco,cocoeval}.py are pure Python; their compiled `_mask` dependency
    comes from $KGDET_REF_BUILD).  None when the reference checkout is absent."""
    if not os.path.isfile(os.path.join(REF_API, 'PythonAPI', 'pycocotools', 'cocoeval.py')):
        return None
    import importlib.util
    import types
    import numpy as np
    if not hasattr(np, 'float'):
        np.float = float          # cocoeval.py:416 uses the alias numpy 1.24 removed
    pkg_dir = os.path.join(REF_API, 'PythonAPI', 'pycocotools')
    pkg = types.ModuleType('pycocotools')
    pkg.__path__ = [pkg_dir]
    sys.modules.setdefault('pycocotools', pkg)
    spec = importlib.util.spec_from_file_location('pycocotools._mask', build_mask_api())
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    sys.modules['pycocotools._mask'] = m
    from pycocotools.coco import COCO
    from pycocotools.cocoeval import COCOeval
    return COCO, COCOeval


def load():
    """Returns (nms_cpu_module, soft_nms_cpu_function) built from the reference, or None."""
    ext = sysconfig.get_config_var('EXT_SUFFIX')
    if available():
        build_nms_cpu()
        build_soft_nms()
    if not os.path.exists(os.path.join(OUT, 'ref_nms_cpu' + ext)):
        return None
    import importlib.util
    import torch  # noqa: F401  (libtorch must be loaded first)
    mods = []
    for name in ('ref_nms_cpu', 'soft_nms_cpu'):
        spec = importlib.util.spec_from_file_location(name, os.path.join(OUT, name + ext))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        mods.append(m)
    return mods[0], mods[1].soft_nms_cpu


if __name__ == '__main__':
    if not available():
        print('reference not present; nothing to build')
    else:
        print(build_nms_cpu())
        print(build_soft_nms())
        print(build_mask_api())
