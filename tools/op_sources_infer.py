"""Which python lines launch the small kernels of a bf16 batch-8 inference step?  torch.profiler with stacks over 3 steps (eager, the
graph replays the same kernels): device time per (op, input shape, innermost kgdet_amd frame).  python tools/op_sources_infer.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from kgdet_amd import build_detector, configs, synthetic
SERIAL = os.environ.get('CONFIG', 'kgdet') == 'serial'      # CONFIG=serial: BASELINE config 5
cfg = configs.reppoints_kp_r50_fpn() if SERIAL else configs.kgdet_r50_fpn()
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().eval()
batch = synthetic.make_batch(8, 'cuda', seed=0)
autocast = torch.autocast('cuda', dtype=torch.bfloat16)
(synthetic.calibrate_scores_serial(model, batch, cfg.test_cfg.score_thr, 0.002, autocast) if SERIAL
 else synthetic.calibrate_scores(model, batch, cfg.test_cfg.score_thr, 0.02, autocast))
def step():
    with torch.no_grad(), autocast:
        return model.simple_test_batch(batch['img'], batch['img_meta'], rescale=True)
for _ in range(4): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
skip = ('aten::conv', 'aten::_conv', 'aten::miopen', 'aten::cudnn', 'aten::mm', 'aten::addmm', 'aten::matmul', 'aten::linear')
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if not e.name.startswith('aten::') or e.self_device_time_total <= 0:
        continue
    frame = next((f for f in (e.stack or []) if 'kgdet_amd' in f), '?')
    frame = frame.split('kgdet_amd/')[-1][:60]
    key = (e.name, str([list(s) for s in (e.input_shapes or []) if s][:1]), frame)
    agg[key][0] += 1
    agg[key][1] += e.self_device_time_total
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]
for (name, shp, frame), (n, t) in rows:
    if name.startswith(skip): continue
    print('%7.1f us  %3d x  %-28s %-28s %s' % (t / 3, n // 3, name, shp, frame))
