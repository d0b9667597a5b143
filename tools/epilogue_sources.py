"""Which convolutions of the bf16 batch-8 inference step still take a separate bias / residual / ReLU pass (backbone._epilogue_ ->
csrc/epilogue.hip bias_act_nhwc)?  One eager step after warm-up: shape, flags, caller, device time.   python tools/epilogue_sources.py"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import build_detector, configs, synthetic, backbone as bb
cfg = configs.kgdet_r50_fpn()
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().eval()
batch = synthetic.make_batch(int(os.environ.get('IMGS', '8')), 'cuda', seed=0)
autocast = torch.autocast('cuda', dtype=torch.bfloat16)
synthetic.calibrate_scores(model, batch, cfg.test_cfg.score_thr, 0.02, autocast)
def step():
    with torch.no_grad(), autocast:
        return model.simple_test_batch(batch['img'], batch['img_meta'], rescale=True)
for _ in range(4): step()
torch.cuda.synchronize()
real = bb._epilogue_
log = []
def spy(y, bias, residual, relu):
    frames = [f for f in traceback.extract_stack()[:-1] if 'kgdet_amd' in f.filename]
    where = ' < '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in frames[-3:][::-1])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = real(y, bias, residual, relu); e1.record()
    log.append((tuple(y.shape), str(y.dtype)[6:], y.is_contiguous(memory_format=torch.channels_last), residual is not None, bool(relu), where, e0, e1))
    return out
bb._epilogue_ = spy
step()
torch.cuda.synchronize()
tot = 0.0
for shp, dt, cl, res, relu, where, e0, e1 in log:
    t = e0.elapsed_time(e1) * 1e3; tot += t
    print('%7.1f us  %-22s %s %s res=%d relu=%d  %s' % (t, shp, dt, 'NHWC' if cl else 'NCHW', res, relu, where))
print('%d passes, %.0f us (event-timed, includes launch gaps)' % (len(log), tot))
print({k: v for k, v in bb._gemm_choice.items()})
