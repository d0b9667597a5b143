"""The KGDet training step as one HIP graph (runner.GraphedTrainStep): eager vs replayed step time, and the same parameters
after the same number of steps.   python tools/graph_train_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import build_detector, configs, synthetic
from kgdet_amd.dist import DistOptimizerHook
from kgdet_amd.runner import GraphedTrainStep, batch_processor
torch.backends.cudnn.benchmark = False
B = int(os.environ.get('IMGS', '2'))


def make():
    cfg = configs.kgdet_r50_fpn()
    torch.manual_seed(0)
    model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-5)
    hook = DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2))
    return model, opt, hook


def timed(fn, n=40):
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3


batch = synthetic.make_batch(B, 'cuda', seed=0)
model, opt, hook = make()
def eager():
    out = batch_processor(model, batch)
    hook.step(model, opt, out['loss'])
for _ in range(8): eager()
print('eager %.2f ms/step' % timed(eager), flush=True)
del model, opt, hook

# same number of steps both ways from the same initial state: 4 (warm-up inside GraphedTrainStep) + 6
m1, o1, h1 = make()
g = GraphedTrainStep(m1, o1, h1, batch, warmup=3)
print('captured', flush=True)
for _ in range(6): out = g.step()
torch.cuda.synchronize()
print('loss after replays %.6f' % float(out['loss']), flush=True)
m2, o2, h2 = make()
for _ in range(3): 
    out2 = batch_processor(m2, batch); h2.step(m2, o2, out2['loss'])
h2._fused.enable_device_schedule(o2)
for _ in range(1 + 6):
    out2 = batch_processor(m2, batch); h2.step(m2, o2, out2['loss'])
torch.cuda.synchronize()
print('loss eager (device schedule) %.6f' % float(out2['loss']), flush=True)
worst = max(float((a - b).abs().max()) for a, b in zip(m1.parameters(), m2.parameters()))
same = all(torch.equal(a, b) for a, b in zip(m1.parameters(), m2.parameters()))
print('parameters after 10 steps: bit-identical %s, max abs difference %.3e' % (same, worst), flush=True)
g.sync_optimizer_state()
print('optimizer step counters', float(next(iter(o1.state.values()))['step']), float(next(iter(o2.state.values()))['step']))
print('graph %.2f ms/step' % timed(g.step), flush=True)
print('graph %.2f ms/step' % timed(g.step), flush=True)
