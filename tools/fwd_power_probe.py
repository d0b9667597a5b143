"""Board power and shader clock while the grouped head-stage forward runs back to back for a few seconds (is the kernel at the
board's power limit?):  [KGDET_LIB=...] python tools/fwd_power_probe.py [seconds]"""
import os, sys, time, glob, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import dcn
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
dev = torch.device('cuda:0')
torch.manual_seed(0)
B, C, H, W = 2, 256, 25, 42
xs = [torch.randn(B, C, H, W, device=dev) for _ in range(2)]
ks = (3, 5, 7)
offs = [torch.randn(B, 2 * k * k, H, W, device=dev) * 2 for k in ks]
ws = [[torch.randn(C, C, k, k, device=dev) * 0.01 for k in ks] for _ in xs]
pr = torch.cuda.get_device_properties(0)
bdf = '%04x:%02x:%02x.0' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
hw = sorted(glob.glob('/sys/bus/pci/devices/%s/hwmon/hwmon*' % bdf))
hw = hw[0] if hw else None
samples, stop = [], threading.Event()
def rd(n):
    try: return float(open(os.path.join(hw, n)).read().split()[0])
    except Exception: return None
def sampler():
    while not stop.is_set():
        p = rd('power1_average') or rd('power1_input'); f = rd('freq1_input')
        if p and f: samples.append((p / 1e6, f / 1e6))
        time.sleep(0.1)
with torch.no_grad():
    for _ in range(50): dcn.deform_conv_cat_multi(xs, offs, ws, [k // 2 for k in ks])
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(200): dcn.deform_conv_cat_multi(xs, offs, ws, [k // 2 for k in ks])
        n += 200
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop.set(); th.join()
us = e0.elapsed_time(e1) / n * 1e3
half = samples[len(samples) // 2:]
print('%.1f us per launch sequence over %d launches; board (second half of %d samples): %.0f W, %.0f MHz' %
      (us, n, len(samples), sum(p for p, _ in half) / max(len(half), 1), sum(f for _, f in half) / max(len(half), 1)))
