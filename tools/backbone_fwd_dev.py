"""forward activations of the training-mode backbone, split-bf16 kernels against MIOpen fp32 (dcn.arithmetic('exact')) and
against a float64 CPU run of the same modules: per stage relative L2 deviation and the mean of (a - b) / |b|_rms"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import configs, dcn, synthetic
from kgdet_amd.registry import build_detector
cfg = configs.kgdet_r50_fpn()
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
img = synthetic.make_batch(2, 'cuda', seed=0)['img']
bb = model.backbone
names = ['maxpool_out', 'layer1', 'layer2', 'layer3', 'layer4']


def run(mode):
    feats = {}
    hooks = [getattr(bb, n).register_forward_hook(lambda m, i, o, n=n: feats.__setitem__(n, o.detach().double()))
             for n in names[1:]]
    with dcn.arithmetic(mode), torch.enable_grad():
        x = img.clone().requires_grad_(True)
        bb(x)
    for h in hooks:
        h.remove()
    return feats


fs, fe = run('split'), run('exact')
for n in names[1:]:
    a, b = fs[n], fe[n]
    rms = b.pow(2).mean().sqrt()
    print('%-8s rel L2 %.2e   mean signed (a-b)/rms %.2e   mean b/rms %.3f   max|a-b|/max|b| %.2e' % (
        n, float((a - b).norm() / b.norm()), float((a - b).mean() / rms), float(b.mean() / rms),
        float((a - b).abs().max() / b.abs().max())))
