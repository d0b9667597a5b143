"""Per-phase cycle counters of dcn_fwd_plane (needs a -DKGDET_TIMING build): python tools/tm.py build_abl/libtm.so [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kgdet_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch, numpy as np
from kgdet_amd import dcn
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device('cuda:0')
torch.manual_seed(0)
C, H, W = 256, 25, 42
xs = [torch.randn(B, C, H, W, device=dev) for _ in range(2)]
ks = (3, 5, 7)
offs = [torch.randn(B, 2 * k * k, H, W, device=dev) * 2 for k in ks]
ws = [[torch.randn(C, C, k, k, device=dev) * 0.01 for k in ks] for _ in xs]
pads = [k // 2 for k in ks]
for prec in ('split', 'bf16'):
    with torch.no_grad(), dcn.forward_precision(prec):
        for _ in range(3):
            dcn.deform_conv_cat_multi(xs, offs, ws, pads)
    torch.cuda.synchronize()
    wsb = list(dcn._workspaces.values())[0]
    G = 256
    off = G * 8 * 256 * 128 * 4
    raw = wsb[off:off + G * 16 * 10 * 8].cpu().numpy().view(np.uint64).reshape(G, 16, 10).astype(np.float64)
    names = ['prologue', 'issue', 'commit', 'produce', 'multiply', 'barrier', 'seg-switch', 'epilogue', 'TOTAL', 'units']
    print(prec, 'mean cycles per wave (all WGs); wave0 / wave4 / max-WG total')
    for i, nme in enumerate(names):
        print('  %-10s %10.0f   cons %10.0f  prod %10.0f  max %10.0f' % (nme, raw[:, :, i].mean(), raw[:, 0, i].mean(), raw[:, 8, i].mean(), raw[:, :, i].max()))
