"""Where do the small elementwise launches of a training step come from?  torch.profiler with stacks over 2 steady steps:
per (op, input shapes) counts and device time for aten::copy_ / fill_ / add / mul / zero_.   python tools/op_sources.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from kgdet_amd import build_detector, configs, synthetic
cfg = configs.kgdet_r50_fpn() if (len(sys.argv) > 1 and sys.argv[1] == "kgdet") else configs.reppoints_kp_r50_fpn()
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
opt = torch.optim.SGD(model.parameters(), lr=1e-5, momentum=0.9, fused=True)
vals = lambda v: v if isinstance(v, (list, tuple)) else [v]
batch = synthetic.make_batch(2, 'cuda', seed=0)
def step():
    losses = model(batch['img'], batch['img_meta'], return_loss=True, gt_bboxes=batch['gt_bboxes'],
                   gt_labels=batch['gt_labels'], gt_keypoints=batch['gt_keypoints'])
    sum(sum(vals(v)) for v in losses.values()).backward()
    opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(4): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(2): step()
    torch.cuda.synchronize()
want = None
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.name.startswith('aten::') and e.device_time_total > 0 and not e.name.startswith(('aten::conv', 'aten::_conv', 'aten::miopen', 'aten::cudnn')):
        key = (e.name, str([list(s) for s in (e.input_shapes or []) if s][:2]))
        agg[key][0] += 1
        agg[key][1] += e.device_time_total
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]
for (name, where), (n, t) in rows:
    print('%7.1f us  %4d x  %-16s %s' % (t / 2, n // 2, name, where))
