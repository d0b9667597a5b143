"""Why does the training step slow down from ~12 ms to ~16 ms over the first ~8 s?  (round 4)

    python tools/steady_probe.py [seconds=20] [steps_per_window=20]

Per window of N steps: host ENQUEUE time (the loop of step() calls returns; nothing waits for the GPU), TOTAL time (after
synchronize), the allocator's counters and the board's sensors from sysfs (shader / memory clock, power, temperatures,
busy percentages).  enqueue ~ total => the host is the limit; enqueue << total => the GPU is.
"""
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('MIOPEN_USER_DB_PATH', os.path.join(ROOT, 'kgdet_amd', 'miopen_db', 'train_fp32_b2'))

import torch  # noqa: E402


def sensors():
    out = {}
    for hw in sorted(glob.glob('/sys/class/drm/card[0-9]*/device/hwmon/hwmon*'))[:1]:
        for f in sorted(os.listdir(hw)):
            if f.endswith('_input') or f.endswith('_average'):
                try:
                    v = open(os.path.join(hw, f)).read().strip()
                    lab = os.path.join(hw, f.split('_')[0] + '_label')
                    name = open(lab).read().strip() if os.path.exists(lab) else ''
                    out['%s(%s)' % (f, name)] = v
                except Exception:
                    pass
    dev = sorted(glob.glob('/sys/class/drm/card[0-9]*/device'))[:1]
    for d in dev:
        for f in ('gpu_busy_percent', 'mem_busy_percent'):
            try:
                out[f] = open(os.path.join(d, f)).read().strip()
            except Exception:
                pass
        for f in ('pp_dpm_sclk', 'pp_dpm_mclk', 'pp_dpm_fclk', 'pp_dpm_socclk'):
            try:
                cur = [l for l in open(os.path.join(d, f)).read().splitlines() if '*' in l]
                out[f] = cur[0].strip() if cur else '?'
            except Exception:
                pass
    return out


def cpu_mhz():
    try:
        v = [float(l.split(':')[1]) for l in open('/proc/cpuinfo') if l.startswith('cpu MHz')]
        return '%.0f..%.0f (n=%d)' % (min(v), max(v), len(v))
    except Exception:
        return '?'


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    from kgdet_amd import configs, synthetic
    from kgdet_amd.dist import DistOptimizerHook
    from kgdet_amd.registry import build_detector
    dev = torch.device('cuda:0')
    cfg = configs.kgdet_r50_fpn()
    torch.manual_seed(0)
    torch.backends.cudnn.benchmark = True
    model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).to(dev)
    batch = synthetic.make_batch(2, dev, seed=0)
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=cfg.optimizer.lr, fused=True)
    hook = DistOptimizerHook(grad_clip=dict(cfg.optimizer_config.grad_clip), overlap=True, bucket_size_mb=32)

    def step():
        losses = model(batch['img'], batch['img_meta'], return_loss=True, gt_bboxes=batch['gt_bboxes'],
                       gt_labels=batch['gt_labels'], gt_keypoints=batch['gt_keypoints'])
        loss = sum(v.float() if torch.is_tensor(v) else sum(x.float() for x in v) for k, v in losses.items() if 'loss' in k)
        hook.step(model, opt, loss)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    # probes that do not depend on the other side: a fixed GPU job timed by events (does the GPU slow down?) and a fixed
    # pure-Python job timed on the host (does the host slow down?)
    pa = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    pb = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    pc = torch.empty(64 << 20, device=dev, dtype=torch.float32)

    def gpu_probe():
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        for _ in range(4):
            torch.mm(pa, pb)
        ev[1].record()
        for _ in range(4):
            pc.mul_(1.0001)
        ev[2].record()
        torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]) / 4, ev[1].elapsed_time(ev[2]) / 4

    def host_probe():
        t = time.perf_counter()
        x = 0
        for i in range(200000):
            x += i * i
        return (time.perf_counter() - t) * 1e3

    import threading
    live = []
    stop = threading.Event()

    def sampler():
        while not stop.is_set():
            s = sensors()
            live.append((s.get('pp_dpm_sclk'), s.get('pp_dpm_fclk'), s.get('pp_dpm_mclk'), s.get('gpu_busy_percent'),
                         [v for k, v in s.items() if k.startswith('power')]))
            stop.wait(0.1)
    threading.Thread(target=sampler, daemon=True).start()
    gpu_probe()
    print('sensors at start:', sensors(), flush=True)
    print('host cpus %d, cpu MHz %s, loadavg %s' % (os.cpu_count(), cpu_mhz(), os.getloadavg()), flush=True)
    t_start = time.time()
    w = 0
    while time.time() - t_start < seconds:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(n):
            step()
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ms = torch.cuda.memory_stats()
        s = sensors()
        gp = gpu_probe()
        hp = host_probe()
        mid = live[len(live) // 2] if live else None
        print('    probes: mm8192 %.3f ms, 256MB mul_ %.3f ms, host loop %.1f ms; in-load samples %d, mid-window %s' % (gp[0], gp[1], hp, len(live), mid), flush=True)
        del live[:]
        print('w%02d t=%5.1fs enqueue %.2f ms/step  total %.2f ms/step  gpu-span %.2f ms/step | reserved %.0f MB segs %d allocs %d retries %d | '
              'sclk %s mclk %s fclk %s P %s busy %s mem %s | cpuMHz %s'
              % (w, time.time() - t_start, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, e0.elapsed_time(e1) / n,
                 torch.cuda.memory_reserved() / 1e6, ms.get('segment.all.current', -1), ms.get('num_device_alloc', -1),
                 ms.get('num_alloc_retries', -1), s.get('pp_dpm_sclk'), s.get('pp_dpm_mclk'), s.get('pp_dpm_fclk'),
                 [v for k, v in s.items() if k.startswith('power')], s.get('gpu_busy_percent'), s.get('mem_busy_percent'),
                 cpu_mhz()), flush=True)
        w += 1
    print('sensors at end:', sensors(), flush=True)
    # the same loop with the GPU kept saturated by a deep queue is what the windows above are; now ONE window where every
    # step is synchronised (host enqueue and GPU execution serialised): its per-step time minus the GPU span = host part
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
        torch.cuda.synchronize()
    print('synchronised steps: %.2f ms/step' % ((time.perf_counter() - t0) / n * 1e3), flush=True)


if __name__ == '__main__':
    main()
