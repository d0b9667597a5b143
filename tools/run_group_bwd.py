"""Forward + backward of one head stage's grouped DeformConv (2 feature maps x 3x3/5x5/7x7 on [B, 256, 25, 42]) a few
times (profiling target for the three backward plane kernels): python tools/run_group_bwd.py [B] [iters] [random|trained|step]
`trained`: the offsets of a converged keypoint-guided head (tests/test_gpu_dcn.py::_keypoint_offsets: tap t of every location of
an image samples one of two key points), the regime the training step is in after a few hundred steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import dcn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device('cuda:0')
torch.manual_seed(0)
C, H, W = 256, 25, 42
ks = (3, 5, 7)
xs = [torch.randn(B, C, H, W, device=dev, requires_grad=True) for _ in range(2)]
offs = [(torch.randn(B, 2 * k * k, H, W, device=dev) * 2) for k in ks]
if len(sys.argv) > 3 and sys.argv[3] == 'trained':
    import numpy as np
    rng = np.random.default_rng(3)
    gy_, gx_ = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    which = (gx_ * 2 // W).clip(0, 1)
    for i, k in enumerate(ks):
        o = offs[i].cpu().numpy()
        for b in range(B):
            for t in range(k * k):
                ky, kx = rng.uniform(1, H - 2, size=2), rng.uniform(1, W - 2, size=2)
                o[b, 2 * t] = ky[which] - (gy_ - k // 2 + t // k) + 0.03 * o[b, 2 * t]
                o[b, 2 * t + 1] = kx[which] - (gx_ - k // 2 + t % k) + 0.03 * o[b, 2 * t + 1]
        offs[i] = torch.from_numpy(o).to(dev)
if len(sys.argv) > 3 and sys.argv[3] == 'step':
    # the offsets of the bench's own steady training state (tools/dump_step_offsets.py: second head stage, step 700): no hot cells,
    # but a quarter of the contributions in cells of 9 .. 64 -- the regime the training step's kernels actually run in
    import numpy as np
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'step_offsets_call1_f16.npz'))
    for i, k in enumerate(ks):
        o = torch.from_numpy(d['call1_k%d' % k].astype('float32'))
        offs[i] = o.repeat((B + 1) // 2, 1, 1, 1)[:B].contiguous().to(dev)
offs = [o.requires_grad_() for o in offs]
ws = [[(torch.randn(C, C, k, k, device=dev) * 0.01).requires_grad_() for k in ks] for _ in xs]
gos = None
for _ in range(iters):
    outs = dcn.deform_conv_cat_multi(xs, offs, ws, [k // 2 for k in ks])
    if gos is None:
        gos = [torch.randn_like(o) for o in outs]
    torch.autograd.backward(outs, gos)
    for t in xs + offs + [w for wl in ws for w in wl]:
        t.grad = None
torch.cuda.synchronize()
