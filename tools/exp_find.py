"""Experiment: bf16 inference with torch.backends.cudnn.benchmark (MIOpen find) on/off."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.backends.cudnn.benchmark = len(sys.argv) > 1 and sys.argv[1] == '1'
from kgdet_amd import build_detector, configs, synthetic
dev = torch.device('cuda:0')
cfg = configs.kgdet_r50_fpn()
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).to(dev).eval()
batch = synthetic.make_batch(8, dev, seed=0)
ac = torch.autocast('cuda', dtype=torch.bfloat16)
def run():
    with torch.no_grad(), ac:
        return model.simple_test_batch(batch['img'], batch['img_meta'], rescale=True)
t0 = time.time()
for _ in range(5): run()
torch.cuda.synchronize(); print('warmup %.1f s' % (time.time() - t0))
t0 = time.time()
for _ in range(20): run()
torch.cuda.synchronize(); print('benchmark=%s: %.2f ms/batch' % (torch.backends.cudnn.benchmark, (time.time() - t0) / 20 * 1e3))
