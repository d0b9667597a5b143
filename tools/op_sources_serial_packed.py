"""Which python lines launch the glue kernels of config 5's PACKED inference chain (the one bench.py captures as a HIP graph)?
torch.profiler with stacks over 3 eager runs of the chain: device time per (op, input shape, innermost kgdet_amd frame).
    python tools/op_sources_serial_packed.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from kgdet_amd import build_detector, configs, synthetic
cfg = configs.reppoints_kp_r50_fpn()
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().eval()
batch = synthetic.make_batch(8, 'cuda', seed=0)
autocast = torch.autocast('cuda', dtype=torch.bfloat16)
synthetic.calibrate_scores_serial(model, batch, cfg.test_cfg.score_thr, 0.002, autocast)
def chain():
    with torch.no_grad(), autocast:
        outs = model.bbox_head(model.extract_feat(batch['img']), batch['img_meta'])
        return model.bbox_head.get_bboxes_packed_tensor(*(outs + (batch['img_meta'], model.test_cfg, True)))
for _ in range(4): chain()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(3): chain()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if not e.name.startswith('aten::') or e.self_device_time_total <= 0:
        continue
    frames = [f for f in (e.stack or []) if 'kgdet_amd' in f]
    frame = (frames[0].split('kgdet_amd/')[-1][:48] if frames else '?')
    key = (e.name, str([list(s) for s in (e.input_shapes or []) if s][:2])[:60], frame)
    agg[key][0] += 1
    agg[key][1] += e.self_device_time_total
for (name, shp, frame), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print('%7.1f us  %3d x  %-26s %-60s %s' % (t / 3, n // 3, name, shp, frame))
