"""what the overlapped gradient exchange costs a step at ONE rank (RCCL's collective is a no-op there): alternating windows of
(a) no exchange, (b) the full reducer, (c) the gradient hooks alone (launch / finish patched out), (d) hooks + pack + collective
without the final p.grad rebinding.  python tools/dist_overhead.py  (run through gpurun)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
os.environ.setdefault('MIOPEN_USER_DB_PATH', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'kgdet_amd', 'miopen_db', 'train_fp32_b2'))
import torch, torch.distributed as dist
from kgdet_amd import configs, synthetic
from kgdet_amd.dist import DistOptimizerHook, OverlappedGradReducer
from kgdet_amd.registry import build_detector
torch.cuda.set_device(0)
dist.init_process_group('nccl')
torch.backends.cudnn.benchmark = True
cfg = configs.kgdet_r50_fpn()
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
batch = synthetic.make_batch(2, 'cuda', seed=0)
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4, fused=True)
hook = DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2), overlap=True, bucket_size_mb=32, force_distributed=True)


def step():
    losses = model(batch['img'], batch['img_meta'], return_loss=True, gt_bboxes=batch['gt_bboxes'], gt_labels=batch['gt_labels'],
                   gt_keypoints=batch['gt_keypoints'])
    hook.step(model, opt, sum(sum(v) for v in losses.values()))


def window(n=30):
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


for _ in range(15):
    step()
orig_launch, orig_finish = OverlappedGradReducer._launch, OverlappedGradReducer.finish


def finish_hooks_only(self):
    if self.buckets is None:
        return orig_finish(self)
    self._reset_counts()


modes = {
    'no exchange': lambda: hook.set_local_only(True),
    'full reducer': lambda: (hook.set_local_only(False), setattr(OverlappedGradReducer, '_launch', orig_launch),
                             setattr(OverlappedGradReducer, 'finish', orig_finish)),
    'hooks only': lambda: (hook.set_local_only(False), setattr(OverlappedGradReducer, '_launch', lambda self, b: self._launched.__setitem__(b, True)),
                           setattr(OverlappedGradReducer, 'finish', finish_hooks_only)),
}
res = {k: [] for k in modes}
for rep in range(4):
    for name, setup in modes.items():
        setup()
        for _ in range(3):
            step()
        res[name].append(window())
for k, v in res.items():
    print('%-14s %s  median %.2f ms' % (k, ['%.2f' % t for t in v], sorted(v)[len(v) // 2]))
