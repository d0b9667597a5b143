"""grouped head-stage forward vs single calls (exact kernel) on the bench shape: python tools/check_group.py [O] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import dcn
O = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
torch.manual_seed(5)
C, H, W = 256, 25, 42
xs = [torch.randn(B, C, H, W, device='cuda') for _ in range(2)]
ks = (3, 5, 7)
offsets = [torch.randn(B, 2 * k * k, H, W, device='cuda') * 2 for k in ks]
weights = [[torch.randn(O, C, k, k, device='cuda') * 0.05 for k in ks] for _ in xs]
pads = [k // 2 for k in ks]
with torch.no_grad():
    print('launching grouped', flush=True)
    outs = dcn.deform_conv_cat_multi(xs, offsets, weights, pads, relu=False)
    torch.cuda.synchronize()
    print('grouped done', flush=True)
    with dcn.forward_precision('exact'):
        for i, x in enumerate(xs):
            ref = torch.cat([dcn.deform_conv(x, offsets[k], weights[i][k], 1, pads[k]) for k in range(3)], 1)
            torch.cuda.synchronize()
            err = (outs[i] - ref).abs().amax(dim=(0, 2, 3)) / ref.abs().max()
            print('map', i, 'max rel err per conv', [float(err[j * O:(j + 1) * O].max()) for j in range(3)], flush=True)
