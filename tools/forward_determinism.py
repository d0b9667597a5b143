import os, sys
sys.path.insert(0, '/root/repo')
import torch
from kgdet_amd import build_detector, configs, synthetic
batch = synthetic.make_batch(2, 'cuda', seed=0)
cfg = configs.reppoints_kp_r50_fpn(soft_nms=True)
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
outs = []
for _ in range(3):
    f = model.extract_feat(batch['img'])
    h = model.bbox_head(f, batch['img_meta']) if 'img_meta' in batch else None
    outs.append(([x.detach().clone() for x in f], [[y.detach().clone() for y in lvl] for lvl in h] if h else None))
for i, (a, b) in enumerate(zip(outs[1][0], outs[2][0])):
    print('feat level', i, tuple(a.shape), 'equal' if torch.equal(a, b) else 'DIFFERS %.2e' % float((a - b).abs().max() / a.abs().max()))
if outs[1][1]:
    for j, (la, lb) in enumerate(zip(outs[1][1], outs[2][1])):
        for i, (a, b) in enumerate(zip(la, lb)):
            if not torch.equal(a, b):
                print('head output', j, 'level', i, tuple(a.shape), 'DIFFERS %.2e' % float((a - b).abs().max() / a.abs().max()))
