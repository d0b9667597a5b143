"""Inference batch: CPU enqueue vs GPU time, and sync points.  python tools/infer_phases.py [B] [bf16]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import build_detector, configs, synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
bf16 = len(sys.argv) > 2 and sys.argv[2] == 'bf16'
dev = torch.device('cuda:0')
cfg = configs.kgdet_r50_fpn()
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).to(dev).eval()
batch = synthetic.make_batch(B, dev, seed=0)
ac = torch.autocast('cuda', dtype=torch.bfloat16, enabled=bf16)
synthetic.calibrate_scores(model, batch, cfg.test_cfg.score_thr, 0.02, ac)
torch.backends.cudnn.benchmark = True
def run():
    with torch.no_grad(), ac:
        return model.simple_test_batch(batch['img'], batch['img_meta'], rescale=True)
for _ in range(5): run()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(10): run()
t1 = time.time(); torch.cuda.synchronize(); t2 = time.time()
print('B=%d bf16=%s: %.2f ms/batch wall, cpu-enqueue %.2f ms/batch' % (B, bf16, (t2 - t0) / 10 * 1e3, (t1 - t0) / 10 * 1e3))
with torch.no_grad(), ac:
    x = model.extract_feat(batch['img']); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10): x = model.extract_feat(batch['img'])
    torch.cuda.synchronize(); t_b = (time.time() - t0) / 10
    t0 = time.time()
    for _ in range(10): outs = model.bbox_head(x, batch['img_meta'])
    torch.cuda.synchronize(); t_h = (time.time() - t0) / 10
    t0 = time.time()
    for _ in range(10): dets = model.bbox_head.get_bboxes(*(outs + (batch['img_meta'], model.test_cfg, True)))
    torch.cuda.synchronize(); t_d = (time.time() - t0) / 10
    t0 = time.time()
    for _ in range(10): res = [model.bbox2result_kp(a, b, c, model.bbox_head.num_classes) for a, b, c in dets]
    torch.cuda.synchronize(); t_r = (time.time() - t0) / 10
print('backbone+neck %.2f ms, head %.2f ms, decode+nms %.2f ms, to-numpy %.2f ms' % (t_b * 1e3, t_h * 1e3, t_d * 1e3, t_r * 1e3))
torch.cuda.set_sync_debug_mode('warn')
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    run()
torch.cuda.set_sync_debug_mode('default')
seen = {}
for x in w:
    key = (os.path.basename(x.filename), x.lineno); seen[key] = seen.get(key, 0) + 1
print('syncs:', seen)
