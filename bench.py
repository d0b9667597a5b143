"""KGDet R50-FPN benchmark on MI355X (contract: see the task statement / DESIGN.md).

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

A step = one data-parallel training iteration of the KGDet detector (configs/kgdet_moment_r50_fpn_1x)
on a seeded synthetic DeepFashion2-shaped batch of 2 images per GPU at 800x1333 (padded 800x1344):
forward, the nine losses, backward, gradient all-reduce over RCCL (overlapped with backward),
grad-clip, Adam step.  Prints ONE JSON line on rank 0.  `value` = images/s over all ranks.

Timing protocol (mmdetection/tools/benchmark.py:84-108 skips its first iterations the same way): after --warmup steps the
first window of --steps steps is kept as `burst_img_s` (the cold board), then the loop PRE-HEATS by wall time -- windows of
--steps steps for >= --preheat-s seconds and until three consecutive windows agree within --steady-tol -- because the step
runs at the board's power limit and the clock settles over the first seconds of load.  `value` is the MEDIAN of the
--windows (7) timed windows AFTER that (every window bracketed by barrier + synchronize, MAX over ranks); all windows,
`steady` (do they agree within 3 %), and the board's power / clock read from sysfs or a rocm-smi child are on the line.

Extra objects on the same line:
  roofline     -- the dominant hand-written launch (DeformConv forward of one head stage: 2 maps x 3x3/5x5/7x7, B=2:
                  45.69 GFLOP, 72.1 MB algorithmic) timed live with HIP events on the stream it runs on;
  cpu_baseline -- the oracle's reference-algorithm deformable conv (materialised im2col + GEMM) forward AND backward of
                  the same head stage plus the reference NMS on 1000 boxes, on the host cores, rank 0, N=1 only,
                  bounded sample, each leg in the unit of its GPU counterpart (TFLOP/s, Mbox/s);
  allreduce    -- (N > 1, or --force-dist on one GPU) bus bandwidth of the gradient exchange and `exposed_ms`: what the
                  overlapped exchange adds to a step.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--imgs-per-gpu', type=int, default=2)
    ap.add_argument('--mode', choices=['train', 'infer'], default='train')
    ap.add_argument('--dtype', choices=['fp32', 'bf16'], default='fp32',
                    help='fp32 (the reference precision) or bf16 autocast for the dense convolutions')
    ap.add_argument('--miopen-find', type=int, choices=[0, 1], default=1,
                    help='torch.backends.cudnn.benchmark (the reference\'s cfg.cudnn_benchmark, '
                         'mmdetection/tools/benchmark.py:53-55): MIOpen measures its solvers per convolution shape '
                         'instead of taking the heuristic pick')
    ap.add_argument('--graph', type=int, choices=[0, 1], default=1,
                    help='inference only: replay the batch as one captured HIP graph (detector.graphed_test_batch)')
    ap.add_argument('--config', choices=['kgdet', 'serial'], default='kgdet',
                    help="kgdet: kgdet_moment_r50_fpn_1x (BASELINE configs 2-4); serial: "
                         "reppoints_moment_serial_r50_fpn_1x-deepfashion2 with soft-NMS (BASELINE config 5)")
    ap.add_argument('--no-inference-leg', action='store_true',
                    help='training mode, 1 GPU: skip the bf16 batch-8 inference measurement (a child process) that '
                         'is reported as the `inference` object of the same JSON line')
    ap.add_argument('--windows', type=int, default=7,
                    help='timed windows of --steps steps each (every window bracketed by barrier + synchronize); `value` is '
                         'the MEDIAN window, min / max / all windows are reported beside it')
    ap.add_argument('--force-dist', action='store_true',
                    help='run the N-rank code path even with --gpus 1: ranks started through a child torch.distributed.run, '
                         'init_process_group(nccl), broadcast, overlapped bucketed all-reduce over RCCL, barriers, '
                         'MAX-reduced time and the `allreduce` object (what a 1-GPU box can prove of the multi-GPU path)')
    ap.add_argument('--preheat-s', type=float, default=8.0,
                    help='training: untimed pre-heat AFTER --warmup, in windows of --steps steps, for at least this many '
                         'seconds of steps AND until three consecutive windows agree within --steady-tol (the board clocks '
                         'down to its power limit over the first seconds of load; `value` is the rate after that ramp)')
    ap.add_argument('--preheat-max-s', type=float, default=30.0, help='give up waiting for a steady state after this long')
    ap.add_argument('--steady-tol', type=float, default=0.02)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--mixed-shapes', action='store_true',
                    help='every second image of the batch has a smaller pad_shape of its own (invalid grid points); without the flag '
                         'the training bench times ONE extra window on such a batch and reports it as `mixed_shapes`')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--pipeline', type=int, choices=[0, 1], default=1,
                    help='inference with --graph 1: overlap the result copy + host unpacking of batch k with batch k + 1 '
                         '(0: each batch is replayed, copied and unpacked before the next starts)')
    ap.add_argument('--graph-train', type=int, choices=[0, 1], default=1,
                    help='training, 1 GPU, KGDet: also measure the step replayed as ONE captured HIP graph '
                         '(runner.GraphedTrainStep) in a child process and report it as `graphed_step`')
    ap.add_argument('--graphed-step-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--no-exact-leg', action='store_true',
                    help='training: skip the short window in plain fp32 arithmetic (`exact_fp32` on the line)')
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks ourselves (the reference's
    tools/dist_train.sh:8-10 does the same with torch.distributed.launch).  This parent never touches the GPU
    (`device_count` does not initialise it on this image); the ranks are CHILD processes and the parent exits with
    the launcher's code -- no exec from a GPU-initialised process."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and os.environ.get('KGDET_BENCH_REHEARSAL') != '1':
        sys.exit('bench.py: --gpus %d requested but this node exposes %d GPU(s); refusing to print a %d-GPU line '
                 'from fewer devices' % (args.gpus, have, args.gpus))
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'),
               OMP_NUM_THREADS=os.environ.get('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // args.gpus))))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def bind_rank_to_its_gpus_numa_node():
    """A rank's loader / launch threads on the CPUs of its GPU's NUMA node (the reference leaves placement to the launcher,
    tools/dist_train.sh:8-10 + mmdet/apis/env.py:26-31; on an 8-GPU MI355X node the GPUs hang off two sockets): reads
    /sys/bus/pci/devices/<bdf>/numa_node of LOCAL_RANK's device from rocm-smi-free sysfs and sets the process affinity to that node's
    cpulist.  Silent no-op where sysfs does not say (containers, one node)."""
    try:
        local = int(os.environ.get('LOCAL_RANK', 0))
        cards = sorted(d for d in os.listdir('/sys/class/drm') if d.startswith('card') and d[4:].isdigit() and
                       os.path.isfile('/sys/class/drm/%s/device/numa_node' % d) and
                       os.path.isfile('/sys/class/drm/%s/device/vendor' % d) and
                       open('/sys/class/drm/%s/device/vendor' % d).read().strip() == '0x1002')
        if local >= len(cards):
            return None
        node = int(open('/sys/class/drm/%s/device/numa_node' % cards[local]).read())
        if node < 0:
            return None
        cpus = set()
        for part in open('/sys/devices/system/node/node%d/cpulist' % node).read().strip().split(','):
            a, _, b = part.partition('-')
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if cpus:
            os.sched_setaffinity(0, cpus)
            return node
    except Exception:
        pass
    return None


ARGS = parse_args() if __name__ == '__main__' else None
NUMA_NODE = None
if ARGS is not None:
    if 'RANK' not in os.environ and (ARGS.gpus > 1 or ARGS.force_dist):
        launch_ranks(ARGS)
    if 'RANK' in os.environ and int(os.environ.get('WORLD_SIZE', 1)) > 1:
        NUMA_NODE = bind_rank_to_its_gpus_numa_node()
    if int(os.environ.get('WORLD_SIZE', 1)) != ARGS.gpus:
        sys.exit('bench.py: --gpus %d but WORLD_SIZE=%s' % (ARGS.gpus, os.environ.get('WORLD_SIZE', '1')))
# MIOpen's measured solver picks for this workload's convolution shapes on MI355X (written by MIOpen itself during
# a first run on the GPU box, see kgdet_amd/miopen_db/README.md): with the records present the find step is a
# lookup instead of minutes of measuring.  One directory per workload; must be set before MIOpen is initialised.
if ARGS is not None and ARGS.miopen_find and 'MIOPEN_USER_DB_PATH' not in os.environ:
    _db = os.path.join(ROOT, 'kgdet_amd', 'miopen_db', '%s%s_%s_b%d' % (
        '' if ARGS.config == 'kgdet' else ARGS.config + '_', ARGS.mode, ARGS.dtype, ARGS.imgs_per_gpu))
    if 'RANK' in os.environ:
        # N ranks must not open ONE user find-db for writing (round-3 review): every rank works on a private copy of the
        # committed records in its own temporary directory (removed at exit); a one-process run keeps the in-tree
        # directory, which is how the records get there in the first place.
        import atexit
        import shutil
        import tempfile
        _priv = tempfile.mkdtemp(prefix='kgdet_miopen_r%s_' % os.environ['RANK'])
        if os.path.isdir(_db):
            shutil.copytree(_db, _priv, dirs_exist_ok=True)
        atexit.register(shutil.rmtree, _priv, True)
        _db = _priv
    os.makedirs(_db, exist_ok=True)
    os.environ['MIOPEN_USER_DB_PATH'] = _db

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

BF16_MFMA_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA (MI355X_MICROARCH.md); AMD's 5 PF figure includes 2:1 sparsity
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
HBM_PEAK_GBS = 8000.0
ROOFLINE_PROFILE = {2: 'profiles/r06_dcn_fwd_plane_group_b2.md', 8: 'profiles/r06_dcn_fwd_plane_group_b8_bf16.md'}
BACKWARD_PROFILE = 'profiles/r06_dcn_bwd_plane_kernels.md'


def profiled_traffic(batch):
    """HBM/fabric bytes per launch sequence of the roofline kernel, from the PMC passes of the committed rocprofv3
    profile (line `traffic_bytes_per_launch: N` = 2 x FETCH_SIZE + WRITE_SIZE of the kernels of one grouped launch,
    corrected as MI355X_MICROARCH.md prescribes).  The counters cannot be collected inside this process; the value is
    a RECORDED measurement of the profiled build, named here by file, or null when no profile is committed."""
    path = os.path.join(ROOT, ROOFLINE_PROFILE.get(batch, ''))
    if not os.path.isfile(path):
        return None, None
    for line in open(path):
        if line.startswith('traffic_bytes_per_launch:'):
            return float(line.split(':', 1)[1].split()[0]), os.path.relpath(path, ROOT)
    return None, None


def kgdet_step_flops(B, H=800, W=1344):
    """Algorithmic flops of ONE KGDet R50-FPN training step on B images of H x W (2 per multiply-add): every convolution's forward,
    + grad_weight of every trainable one, + grad_input wherever something trainable lies upstream (frozen_stages = 1: stem and
    layer 1 are forward only, and layer 2's first conv1 / downsample need no grad_input), deformable convolutions x 3 (forward,
    the column-gradient GEMM W^T grad_out ONCE -- the reference shares it between col2im and col2im_coord,
    deform_conv_cuda.cpp:329-371; this build multiplies it in both kernels --, grad_weight).  Returns (total, by family)."""
    conv = lambda cin, cout, k, px: 2.0 * cin * cout * k * k * px
    px = lambda s: (H // s) * (W // s)
    fwd = frozen = no_dgrad = 0.0
    frozen += conv(3, 64, 7, px(2))                                   # stem
    cin = 64
    for li, (planes, blocks, s_in, s_out) in enumerate([(64, 3, 4, 4), (128, 4, 4, 8), (256, 6, 8, 16), (512, 3, 16, 32)]):
        for b in range(blocks):
            first = b == 0
            layer = [conv(cin, planes, 1, px(s_in if first else s_out)), conv(planes, planes, 3, px(s_out)),
                     conv(planes, planes * 4, 1, px(s_out))] + ([conv(cin, planes * 4, 1, px(s_out))] if first else [])
            if li == 0:
                frozen += sum(layer)
            else:
                fwd += sum(layer)
                if li == 1 and first:
                    no_dgrad += layer[0] + layer[3]
            cin = planes * 4
    backbone = frozen + 3 * fwd - no_dgrad
    p32 = px(32)
    fpn = 3 * (conv(2048, 256, 1, p32) + conv(256, 256, 3, p32))      # FPN2 select_out=[2]: lateral_convs[2] + fpn_convs[2]
    dense = 6 * conv(256, 256, 3, p32) + 2 * conv(256, 256, 3, p32) + conv(256, 13, 1, p32) + conv(256, 588, 1, p32) + \
        conv(588, 166, 1, p32) + 2 * (conv(768, 13, 1, p32) + conv(768, 588, 1, p32) + conv(588, 166, 1, p32))
    dcn = 2 * 2 * sum(conv(256, 256, k, p32) for k in (3, 5, 7))      # two deformable stages x (cls, keypoint) branch
    head = 3 * dense + 3 * dcn
    fam = {'backbone': B * backbone, 'fpn': B * fpn, 'head_dense': B * 3 * dense, 'head_deformable': B * 3 * dcn}
    return B * (backbone + fpn + head), fam


# algorithmic flops of ONE training step per GPU: (config, images per GPU) -> flops
STEP_FLOPS = {('kgdet', 2): kgdet_step_flops(2)[0]}
STEP_PROFILE = 'profiles/r06_train_step_fp32_steady.md'
STEP_F64 = 'profiles/r06_step_vs_f64.json'


def step_families():
    """kernel time per family of the committed steady-state profile (lines `family_us <name>: <us per step>`), or nothing"""
    path = os.path.join(ROOT, STEP_PROFILE)
    res = {}
    if os.path.isfile(path):
        for line in open(path):
            if line.startswith('family_us '):
                k, v = line[len('family_us '):].split(':', 1)
                res[k.strip()] = float(v.split()[0])
        res['source'] = STEP_PROFILE
    return res or None


def value_error_vs_f64():
    """the split-operand step's measured deviation from the float64 golden of the same step (tools/step_vs_f64.py --json on the GPU,
    committed): worst relative deviation of a module group's gradient norm and of the nine losses"""
    path = os.path.join(ROOT, STEP_F64)
    if not os.path.isfile(path):
        return None
    try:
        d = json.load(open(path))
        d['source'] = STEP_F64
        return d
    except Exception:
        return None


def dcn_roofline(device, batch=2, precision='split', iters=100):
    """DeformConv forward of ONE KGDet head stage: the grouped launch of 2 feature maps x (3x3, 5x5, 7x7) on
    [batch, 256, 25, 42] (kgdet_deform_conv_forward_grouped: dcn_build_taps + dcn_fwd_plane + dcn_fwd_fixup),
    HIP-event timing on the launch stream.  'split': the products are bf16 MFMAs on a hi/lo split of both fp32
    operands (3 MFMAs per fp32-accurate multiply), so the MFMA roof is the dense bf16 peak / 3; 'bf16' (autocast
    inference): one bf16 MFMA per multiply, roof = the dense bf16 peak."""
    from kgdet_amd import dcn
    g = torch.Generator(device='cpu').manual_seed(0)
    B, C, H, W = batch, 256, 25, 42
    ks = (3, 5, 7)
    xs = [torch.randn(B, C, H, W, generator=g).to(device) for _ in range(2)]
    offs = [(torch.randn(B, 2 * k * k, H, W, generator=g) * 2).to(device) for k in ks]
    ws = [[(torch.randn(C, C, k, k, generator=g) * 0.01).to(device) for k in ks] for _ in xs]
    pads = [k // 2 for k in ks]
    stream = torch.cuda.current_stream()
    with torch.no_grad(), dcn.forward_precision(precision):     # weight images are packed once (inference path)
        for _ in range(30):    # weight images packed, clocks up (a cold burst of 20 launches reads ~8 % slower)
            dcn.deform_conv_cat_multi(xs, offs, ws, pads)
        # five event-bracketed runs of `iters` launches, median run: a host hiccup between two launches (the
        # launch sequence is enqueued from Python) would otherwise be billed to the kernel
        times = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(iters):
                dcn.deform_conv_cat_multi(xs, offs, ws, pads)
            e1.record(stream)
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / iters * 1e-3)
    t = sorted(times)[len(times) // 2]
    flops = sum(2.0 * C * C * k * k * B * H * W for k in ks) * len(xs)
    byts = sum(4.0 * (2 * B * C * H * W + 2 * B * k * k * H * W + C * C * k * k) for k in ks) * len(xs)
    ach = flops / t / 1e12
    parts = 3.0 if precision == 'split' else 1.0
    peak = BF16_MFMA_PEAK_TFLOPS / parts
    traffic, traffic_src = profiled_traffic(B)
    return dict(bound='mfma', kernel='dcn_build_taps+dcn_fwd_plane<%d>+dcn_fwd_fixup (one head stage: 2 maps x 3x3/5x5/7x7, '
                                     'B=%d, 256ch, 25x42)' % (2 if precision == 'split' else 1, B),
                achieved=round(ach, 2), peak=round(peak, 1), unit='TFLOP/s', frac=round(ach / peak, 4),
                peak_basis='dense bf16 MFMA 2500 TFLOP/s / 3 products per fp32-accurate multiply (hi/lo split)'
                if precision == 'split' else 'dense bf16 MFMA 2500 TFLOP/s (operands rounded to bf16 once)',
                frac_of_f32_mfma_peak=round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
                traffic=traffic, traffic_source=traffic_src, algorithmic_bytes=byts, algorithmic_flops=flops,
                launch_us=round(t * 1e6, 1), hbm_achieved_GBs=round(byts / t / 1e9, 1),
                hbm_frac=round(byts / t / 1e9 / HBM_PEAK_GBS, 4))


def dcn_backward_live(device, iters=30):
    """Backward of the same head stage (grad_input + grad_offset + grad_weight of the six convs: 3 x 45.69 GFLOP), HIP
    events on the launch stream: (forward + backward) minus forward, median of five runs."""
    from kgdet_amd import dcn
    g = torch.Generator(device='cpu').manual_seed(0)
    B, C, H, W = 2, 256, 25, 42
    ks = (3, 5, 7)
    xs = [torch.randn(B, C, H, W, generator=g).to(device).requires_grad_() for _ in range(2)]
    offs = [(torch.randn(B, 2 * k * k, H, W, generator=g) * 2).to(device).requires_grad_() for k in ks]
    ws = [[(torch.randn(C, C, k, k, generator=g) * 0.01).to(device).requires_grad_() for k in ks] for _ in xs]
    pads = [k // 2 for k in ks]
    leaves = xs + offs + [w for wl in ws for w in wl]
    stream = torch.cuda.current_stream()
    gos = None

    def run(backward):
        nonlocal gos
        outs = dcn.deform_conv_cat_multi(xs, offs, ws, pads)
        if gos is None:
            gos = [torch.randn_like(o) for o in outs]
        if backward:
            torch.autograd.backward(outs, gos)
            for t in leaves:
                t.grad = None

    med = {}
    for backward in (False, True):
        for _ in range(10):
            run(backward)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(iters):
                run(backward)
            e1.record(stream)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / iters * 1e-3)
        med[backward] = sorted(ts)[2]
    return med[True] - med[False]


def _trained_offsets(offs, ks, B, H, W):
    """the offsets of a converged keypoint-guided head (tests/test_gpu_dcn.py::_keypoint_offsets): tap t of every location of an
    image samples one of two key points -- the regime the training step is in after a few hundred steps"""
    import numpy as np
    rng = np.random.default_rng(3)
    gy, gx = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    which = (gx * 2 // W).clip(0, 1)
    out = []
    for o, k in zip(offs, ks):
        a = o.cpu().numpy().copy()
        for b in range(B):
            for t in range(k * k):
                ky, kx = rng.uniform(1, H - 2, size=2), rng.uniform(1, W - 2, size=2)
                a[b, 2 * t] = ky[which] - (gy - k // 2 + t // k) + 0.03 * a[b, 2 * t]
                a[b, 2 * t + 1] = kx[which] - (gx - k // 2 + t % k) + 0.03 * a[b, 2 * t + 1]
        out.append(torch.from_numpy(a).to(o.device))
    return out


STEP_OFFSETS = 'tools/data/step_offsets_call1_f16.npz'


def _step_offsets(device, ks):
    """the offsets the benched training step itself produces in its steady state (second head stage, step 700 of this bench's model
    on its synthetic batch; tools/dump_step_offsets.py, stored in fp16): no cell collects more than 64 contributions, but a quarter
    of all contributions sit in cells of 9 .. 64 -- neither the random nor the converged-key-point regime"""
    import numpy as np
    d = np.load(os.path.join(ROOT, STEP_OFFSETS))
    return [torch.from_numpy(d['call1_k%d' % k].astype('float32')).to(device) for k in ks]


def dcn_backward_products_live(device, iters=30, regime='random'):
    """The three backward products of the same head stage, each timed alone: (forward + ONE product) minus forward, HIP events
    on the launch stream, median of five runs.  grad_weight: only the weights require gradients; grad_input / grad_offset: the
    grouped call's two phases one at a time (KGDET_OPT_BWD_PHASE, a measurement switch of the C ABI).  ``regime``: 'random'
    N(0, 2^2)-pixel offsets, 'trained' (``_trained_offsets``) or 'step' (``_step_offsets``: the benched step's own)."""
    from kgdet_amd import _lib, dcn
    g = torch.Generator(device='cpu').manual_seed(0)
    B, C, H, W = 2, 256, 25, 42
    ks = (3, 5, 7)
    xs = [torch.randn(B, C, H, W, generator=g).to(device) for _ in range(2)]
    offs = [(torch.randn(B, 2 * k * k, H, W, generator=g) * 2).to(device) for k in ks]
    ws = [[(torch.randn(C, C, k, k, generator=g) * 0.01).to(device) for k in ks] for _ in xs]
    if regime == 'trained':
        offs = _trained_offsets(offs, ks, B, H, W)
    elif regime == 'step':
        offs = _step_offsets(device, ks)
    pads = [k // 2 for k in ks]
    leaves_io, leaves_w = xs + offs, [w for wl in ws for w in wl]
    stream = torch.cuda.current_stream()
    gos = [None]

    def run(which):
        # ('forward': the forward of a training step -- gradients required, so the weights are re-packed and tensors saved)
        for t in leaves_io:
            t.requires_grad_(which in ('grad_input', 'grad_offset'))
        for t in leaves_w:
            t.requires_grad_(which in ('grad_weight', 'forward'))
        outs = dcn.deform_conv_cat_multi(xs, offs, ws, pads)
        if gos[0] is None:
            gos[0] = [torch.randn_like(o) for o in outs]
        if which != 'forward':
            torch.autograd.backward(outs, gos[0])
            for t in leaves_io + leaves_w:
                t.grad = None

    replayed = [True]

    def timed(which):
        """the launches of one call captured as a HIP graph and replayed back to back (what the GPU-bound training step looks like
        to the device: a Python loop of forward + one product enqueues ~250 us per iteration, more than the product itself, and the
        gaps were billed to the kernels); eager enqueue if the capture fails"""
        for _ in range(10):
            run(which)
        torch.cuda.synchronize()
        graph = None
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                run(which)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    run(which)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
        except Exception:
            graph = None
            replayed[0] = False
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(iters):
                if graph is not None:
                    graph.replay()
                else:
                    run(which)
            e1.record(stream)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / iters * 1e-3)
        return sorted(ts)[2]

    flops = sum(2.0 * C * C * k * k * B * H * W for k in ks) * 2
    out = {}
    dcn.MEASUREMENT = True          # (the grouped backward reports a phase-only call as KGDET_E_PARTIAL; accepted here only)
    try:
        base = timed('forward')
        for which, phase in (('grad_weight', 0), ('grad_input', 1), ('grad_offset', 2)):
            _lib.check(_lib.lib().kgdet_set_option(3, phase), 'kgdet_set_option')
            t = max(timed(which) - base, 1e-9)
            out[which] = {'us': round(t * 1e6, 1), 'achieved': round(flops / t / 1e12, 2),
                          'frac': round(flops / t / 1e12 / (BF16_MFMA_PEAK_TFLOPS / 3.0), 4)}
    finally:
        _lib.lib().kgdet_set_option(3, 0)
        dcn.MEASUREMENT = False
    out['forward_with_pack_us'] = round(base * 1e6, 1)
    out['offsets'] = regime
    out['timed_as'] = 'hip graph replay of one call\'s launches, back to back' if replayed[0] else 'eager enqueue'
    out['note'] = ('one head stage (2 maps x 3x3/5x5/7x7, B=2, 45.69 GFLOP per product), each product = (forward + that product) - '
                   'forward, HIP events, median of five runs of %d; frac against the forward\'s 833 TFLOP/s split-bf16 roof; a '
                   'product includes its builder kernels (inverse records / long-cell sums / grad_out images) and the '
                   'ReLU-backward pass; forward_with_pack_us = the forward as a TRAINING step runs it (weights re-packed per call: '
                   'dcn_pack_weight_all_multi inside).  Kernels inside each live figure -- grad_input: dcn_bwd_input_prepare (inverse records '
                   '+ pixel-major grad_out) + dcn_hot_gemm + dcn_inv_medium_sums + dcn_bwd_input_plane<2> + dcn_fwd_fixup_static; grad_offset: '
                   'dcn_build_grad_taps + dcn_bwd_offset_pair + dcn_bwd_offset_plane_fixup; grad_weight: dcn_build_taps + dcn_pack_grad_out '
                   '+ dcn_bwd_weight_os; each + the autograd node\'s ReLU-backward pass on grad_out.  The profile\'s per-kernel averages '
                   '(profiles/) exclude the launch gaps and that pass' % iters)
    tr = profiled_traffic_backward({'trained': 'trained_', 'step': 'step_'}.get(regime, ''))
    for k in ('grad_input', 'grad_offset', 'grad_weight'):
        if k in out and k in tr:
            out[k]['traffic'] = tr[k]
    out['traffic_source'] = tr.get('source')
    return out


def profiled_traffic_backward(prefix=''):
    """per-product fabric bytes of the committed backward profile (lines `traffic_bytes_[trained_]<product>: N`), or nothing"""
    path = os.path.join(ROOT, BACKWARD_PROFILE)
    res = {}
    if os.path.isfile(path):
        for line in open(path):
            if line.startswith('traffic_bytes_' + prefix + 'grad_'):
                key, val = line.split(':', 1)
                res[key[len('traffic_bytes_' + prefix):]] = float(val.split()[0])
        res['source'] = BACKWARD_PROFILE
    return res


def nms_live(device, dets, segments, iters=50):
    """`segments` copies of the box set as ONE batched launch (kgdet_nms_batched: a workgroup per segment) -- the shape
    the inference batch produces (8 images x 13 classes); HIP events around the launches."""
    import numpy as np
    from kgdet_amd.nms import nms_batched
    n = dets.shape[0]
    d = torch.from_numpy(np.tile(dets, (segments, 1))).to(device)
    offs = torch.arange(0, (segments + 1) * n, n, dtype=torch.int64, device=device)
    for _ in range(5):
        nms_batched(d, offs, 0.5, max_seg_len=n)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        nms_batched(d, offs, 0.5, max_seg_len=n)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def cpu_baseline(device=None, gpu_forward_s=None):
    """SURVEY 8d protocol, in the roofline row's own unit on both sides: the oracle's reference algorithm (materialised
    im2col + BLAS GEMM, deform_conv_cuda.cpp:151-484 restated on the CPU) for ONE KGDet head stage at B=2 -- the six
    DeformConv calls (2 maps x 3x3/5x5/7x7 on [2,256,25,42]) of the roofline launch -- forward (45.69 GFLOP) and backward
    (grad_input + grad_offset + grad_weight, 3 x 45.69 GFLOP), plus the reference's nms_cpu algorithm on 1000 boxes; same
    seeded inputs, bounded sample (~20 s), BLAS threads stated.  The GPU's live numbers for the same legs sit beside them."""
    import numpy as np
    import oracle
    # BLAS threads PINNED (round-3 review: 64 OpenBLAS threads roaming over 256 shared host CPUs gave 5x5 = 0.95 x 7x7): at
    # most 16 threads for the whole baseline, stated on the line; every leg reports its MINIMUM beside the median
    pin = max(1, min(16, os.cpu_count() or 1))
    limiter = None
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        limiter = threadpool_limits(limits=pin)
        blas = [(i.get('internal_api'), i.get('num_threads')) for i in threadpool_info() if i.get('user_api') == 'blas']
    except Exception:
        blas = []
    threads = max([n for _, n in blas], default=pin)
    rng = np.random.default_rng(0)
    B, C, H, W = 2, 256, 25, 42
    x = rng.normal(size=(B, C, H, W)).astype(np.float32)
    go = rng.normal(size=(B, C, H, W)).astype(np.float32)

    mins = {}

    def median_time(fn, warm, reps, tag=None):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        if tag is not None:
            mins[tag] = min(ts)
        return sorted(ts)[len(ts) // 2]

    fwd, bwd = {}, {}
    for k in (3, 5, 7):
        off = (rng.normal(size=(B, 2 * k * k, H, W)) * 2).astype(np.float32)
        w = (rng.normal(size=(C, C, k, k)) * 0.01).astype(np.float32)
        fwd[k] = median_time(lambda: oracle.deform_conv_forward(x, off, w, 1, k // 2, 1), 3, 9, 'fwd%d' % k)
        bwd[k] = median_time(lambda: oracle.deform_conv_backward(x, off, w, go, 1, k // 2, 1), 1, 5, 'bwd%d' % k)
    flops = sum(2.0 * C * C * k * k * B * H * W for k in (3, 5, 7)) * 2          # one head stage
    stage_fwd_s, stage_bwd_s = 2 * sum(fwd.values()), 2 * sum(bwd.values())
    brng = np.random.default_rng(5)              # 1000 boxes in 40 clusters: overlapping neighbours, ~150 survivors
    ctr = brng.uniform(100, 1200, size=(40, 2))[brng.integers(0, 40, size=1000)] + brng.normal(0, 12, size=(1000, 2))
    wh = brng.uniform(40, 160, size=(1000, 2))
    boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2, brng.uniform(0.05, 1, size=(1000, 1))], 1).astype(np.float32)
    nms_s = median_time(lambda: oracle.nms(boxes, 0.5), 3, 50)
    legs = {
        'dcn_forward': {'cpu': round(flops / stage_fwd_s / 1e12, 4), 'unit': 'TFLOP/s (one head stage, 45.69 GFLOP)',
                        'cpu_ms': round(stage_fwd_s * 1e3, 1)},
        'dcn_backward': {'cpu': round(3 * flops / stage_bwd_s / 1e12, 4),
                         'unit': 'TFLOP/s (one head stage, grad_input + grad_offset + grad_weight = 137.07 GFLOP)',
                         'cpu_ms': round(stage_bwd_s * 1e3, 1)},
        'nms_1000': {'cpu': round(1000 / nms_s / 1e6, 3),
                     'unit': 'Mbox/s (nms_cpu.cpp algorithm, segments of 1000 boxes, thr 0.5; CPU: one segment after the '
                             'other on one core; GPU: the 104 segments of an inference batch -- 8 images x 13 classes -- '
                             'as one batched launch)',
                     'cpu_ms': round(nms_s * 1e3, 3), 'cpu_ms_104_segments': round(104 * nms_s * 1e3, 2)},
    }
    legs['dcn_forward']['cpu_ms_min'] = round(2 * sum(mins['fwd%d' % k] for k in (3, 5, 7)) * 1e3, 1)
    legs['dcn_backward']['cpu_ms_min'] = round(2 * sum(mins['bwd%d' % k] for k in (3, 5, 7)) * 1e3, 1)
    if limiter is not None:
        limiter.restore_original_limits()
    if device is not None:
        if gpu_forward_s:
            legs['dcn_forward'].update(gpu=round(flops / gpu_forward_s / 1e12, 2), gpu_ms=round(gpu_forward_s * 1e3, 4))
        tb = dcn_backward_live(device)
        legs['dcn_backward'].update(gpu=round(3 * flops / tb / 1e12, 2), gpu_ms=round(tb * 1e3, 4))
        tn = nms_live(device, boxes, 104)
        legs['nms_1000'].update(gpu=round(104 * 1000 / tn / 1e6, 3), gpu_ms_104_segments=round(tn * 1e3, 4))
    return dict(value=legs['dcn_forward']['cpu'], unit='TFLOP/s (DeformConv forward of one KGDet head stage: the roofline '
                'row\'s launch and unit)', cores=threads, host_cpus=os.cpu_count(), kind='port', blas=blas, legs=legs,
                sample='oracle im2col+GEMM on [2,256,25,42]: forward 3x3/5x5/7x7 3 warm-up + median of 9 each '
                       '(%.1f / %.1f / %.1f ms), backward 1 warm-up + median of 5 each (%.1f / %.1f / %.1f ms), x2 maps = one '
                       'head stage; oracle nms 1000 boxes median of 50'
                       % tuple([fwd[k] * 1e3 for k in (3, 5, 7)] + [bwd[k] * 1e3 for k in (3, 5, 7)]))


class BoardSampler:
    """Board power and shader clock during the timed windows, sampled by a host thread every 0.25 s: sysfs first
    (amdgpu hwmon `power1_average` / `power1_input` in microwatts, `freq1_input` in Hz -- plain file reads, no GPU
    call), else a `rocm-smi --showpower --showclocks --json` CHILD process per tick (started, never exec'ed from this
    GPU-initialised process).  Reported on the line, never used for control."""

    def __init__(self, index=0):
        import glob
        self.samples = []
        self._stop = None
        self._thread = None
        self.source = None
        # the hwmon of THIS process's device, found through its PCI address (the host's sysfs lists every board of the node,
        # the container sees one of them); fall back to the i-th amdgpu hwmon in card order
        self.hwmon = None
        try:
            pr = torch.cuda.get_device_properties(index)
            bdf = '%04x:%02x:%02x.0' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            hit = sorted(glob.glob('/sys/bus/pci/devices/%s/hwmon/hwmon*' % bdf))
            if hit:
                self.hwmon = hit[0]
        except Exception:
            pass
        if self.hwmon is None:
            cards = sorted(glob.glob('/sys/class/drm/card[0-9]*/device/hwmon/hwmon*'))
            hw = [h for h in cards if any(os.path.exists(os.path.join(h, f)) for f in ('power1_average', 'power1_input'))]
            self.hwmon = hw[index] if index < len(hw) else None
        if self.hwmon:
            self.source = 'sysfs:' + self.hwmon
        else:
            import shutil
            self.smi = shutil.which('rocm-smi') or ('/opt/rocm/bin/rocm-smi' if os.path.exists('/opt/rocm/bin/rocm-smi') else None)
            self.index = index
            if self.smi:
                self.source = 'rocm-smi child'

    def _read_sysfs(self):
        def rd(name):
            try:
                return float(open(os.path.join(self.hwmon, name)).read().split()[0])
            except Exception:
                return None
        p = rd('power1_average')
        if p is None:
            p = rd('power1_input')
        f = rd('freq1_input')
        return (p / 1e6 if p else None, f / 1e6 if f else None)

    def _read_smi(self):
        import subprocess
        try:
            r = subprocess.run([self.smi, '-d', str(self.index), '--showpower', '--showclocks', '--json'],
                               stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=10)
            d = list(json.loads(r.stdout.decode()).values())[0]
            pw = [float(v) for k, v in d.items() if 'power' in k.lower() and str(v).replace('.', '', 1).isdigit()]
            ck = [float(str(v).strip('()Mhz ')) for k, v in d.items() if 'sclk clock speed' in k.lower()]
            return (pw[0] if pw else None, ck[0] if ck else None)
        except Exception:
            return (None, None)

    def start(self):
        if self.source is None:
            return
        import threading
        self._stop = threading.Event()
        read = self._read_sysfs if self.hwmon else self._read_smi

        def loop():
            while not self._stop.is_set():
                self.samples.append(read())
                self._stop.wait(0.25)
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()

    def stop(self):
        if self._thread is not None:
            self._stop.set()
            self._thread.join(timeout=15)

    def summary(self):
        pw = [p for p, _ in self.samples if p]
        ck = [c for _, c in self.samples if c]
        if not pw and not ck:
            return {'source': self.source, 'samples': len(self.samples), 'note': 'board power / clock not readable here'}
        return {'source': self.source, 'samples': len(self.samples),
                'power_W_mean': round(sum(pw) / len(pw), 1) if pw else None, 'power_W_max': round(max(pw), 1) if pw else None,
                'sclk_MHz_mean': round(sum(ck) / len(ck), 1) if ck else None, 'sclk_MHz_min': round(min(ck), 1) if ck else None}


def allreduce_busbw(device, world, numel=52250071, iters=10):
    """Bus bandwidth of the gradient exchange on its own: ONE fp32 buffer of the step's whole payload
    (52 250 071 gradient values = 209.0 MB), ring convention busbw = 2(N-1)/N * bytes / t."""
    buf = torch.zeros(numel, dtype=torch.float32, device=device)
    for _ in range(3):
        dist.all_reduce(buf)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.time()
    for _ in range(iters):
        dist.all_reduce(buf)
    torch.cuda.synchronize()
    t = torch.tensor([(time.time() - t0) / iters], device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    t = float(t.item())
    byts = numel * 4.0
    return dict(payload_MB=round(byts / 1e6, 1), ms=round(t * 1e3, 3), algbw_GBs=round(byts / t / 1e9, 1),
                busbw_GBs=round(2.0 * (world - 1) / world * byts / t / 1e9, 1),
                note='standalone (not overlapped) all-reduce of the full gradient payload over RCCL/xGMI; in the '
                     'step it runs in 32 MB buckets on a side stream under backward')


def inference_leg():
    """BASELINE's second metric on the same line: inference img/s, bf16, batch 8 at 800x1333 (config 2), hipGraph
    replay, decode + NMS at work -- measured by a CHILD process (MIOpen's per-workload find database is chosen by an
    environment variable that must be set before MIOpen initialises; the child is started, never exec'ed)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--mode', 'infer', '--dtype', 'bf16', '--imgs-per-gpu', '8',
           '--steps', '30', '--warmup', '5', '--no-cpu-baseline']
    env = {k: v for k, v in os.environ.items() if k != 'MIOPEN_USER_DB_PATH'}
    try:
        res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        line = [l for l in res.stdout.decode().splitlines() if l.startswith('{')][-1]
        d = json.loads(line)
    except Exception as e:      # the training number stands on its own; report the failure instead of hiding it
        return {'error': '%s: %s' % (type(e).__name__, e)}
    return {k: d[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'dtype', 'config', 'roofline', 'with_h2d', 'batch_at_a_time') if k in d}


def graphed_step_leg(args):
    """The same training step replayed as ONE captured HIP graph (runner.GraphedTrainStep), measured by a CHILD process with
    the same timing protocol (pre-heat by wall time, median of --windows windows of --steps steps): a capture failure cannot
    take the eager line down with it."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--graphed-step-child', '--steps', str(args.steps), '--warmup',
           str(args.warmup), '--windows', str(args.windows), '--preheat-s', str(min(args.preheat_s, 4.0)),
           '--imgs-per-gpu', str(args.imgs_per_gpu), '--config', args.config, '--no-cpu-baseline', '--no-roofline',
           '--no-inference-leg', '--no-exact-leg']
    try:
        res = subprocess.run(cmd, env=dict(os.environ), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        line = [l for l in res.stdout.decode().splitlines() if l.startswith('{')][-1]
        return json.loads(line)
    except Exception as e:
        return {'error': '%s: %s' % (type(e).__name__, e)}


def main():
    args = ARGS
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    assert torch.cuda.is_available(), 'bench.py needs a GPU: the HIP path is the product, there is no fallback'
    # KGDET_BENCH_REHEARSAL=1: a dry run of the N-rank FLOW on a box with fewer GPUs -- every rank on device 0, gloo instead of
    # RCCL (two ranks cannot share a device under RCCL).  Exercises what a scaling run executes (rank start-up, broadcast, the
    # reducer's collectives in the timed windows, the barriers, rank 0's extra legs) so that a hang shows up here; the line it
    # prints says `rehearsal: true` and is NOT a measurement.
    rehearsal = os.environ.get('KGDET_BENCH_REHEARSAL') == '1'
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    dist_on = world > 1 or args.force_dist      # the N-rank code path (forced: a one-rank RCCL group)
    if dist_on:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='gloo' if rehearsal else 'nccl')

    import kgdet_amd
    from kgdet_amd import configs, synthetic
    from kgdet_amd.dist import DistOptimizerHook
    from kgdet_amd.registry import build_detector

    cfg = configs.kgdet_r50_fpn() if args.config == 'kgdet' else configs.reppoints_kp_r50_fpn(soft_nms=True)
    if args.config == 'serial':
        cfg['optimizer'] = dict(type='SGD', lr=5e-3)      # (the serial config's own optimizer: SGD 5e-3, momentum 0.9, wd 1e-4)
        cfg['optimizer_config'] = dict(grad_clip=dict(max_norm=35, norm_type=2))
    torch.manual_seed(0)
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).to(device)
    if dist_on:  # same initial weights everywhere (the reference broadcasts once at start-up)
        for p in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(p.data, 0)
    batch = synthetic.make_batch(args.imgs_per_gpu, device, seed=rank, mixed_shapes=args.mixed_shapes)
    autocast = torch.autocast('cuda', dtype=torch.bfloat16, enabled=args.dtype == 'bf16')

    if args.mode == 'train':
        model.train()
        params = [p for p in model.parameters() if p.requires_grad]
        # fused: one multi-tensor launch per step instead of ~10 (CPU-bound tail)
        optimizer = torch.optim.Adam(params, lr=cfg.optimizer.lr, fused=True) if args.config == 'kgdet' else \
            torch.optim.SGD(params, lr=cfg.optimizer.lr, momentum=0.9, weight_decay=0.0001, fused=True)
        hook = DistOptimizerHook(grad_clip=dict(cfg.optimizer_config.grad_clip), overlap=True, bucket_size_mb=32,
                                 force_distributed=args.force_dist)

        def step():
            with autocast:     # (`batch` is read at call time: the mixed-shapes window rebinds it)
                losses = model(batch['img'], batch['img_meta'], return_loss=True, gt_bboxes=batch['gt_bboxes'],
                               gt_labels=batch['gt_labels'], gt_keypoints=batch['gt_keypoints'])
            loss = sum(v.float() if torch.is_tensor(v) else sum(x.float() for x in v) for k, v in losses.items()
                       if 'loss' in k)
            hook.step(model, optimizer, loss)
            return loss

        if args.graphed_step_child:
            # the child of graphed_step_leg: capture, pre-heat, time, print, leave
            from kgdet_amd.runner import GraphedTrainStep
            gstep = GraphedTrainStep(model, optimizer, hook, batch, warmup=max(3, args.warmup))

            def gwindow():
                torch.cuda.synchronize()
                t0 = time.time()
                for _ in range(args.steps):
                    gstep.step()
                torch.cuda.synchronize()
                return time.time() - t0
            spent, hist = 0.0, []
            while spent < args.preheat_s or len(hist) < 3:
                hist.append(gwindow())
                spent += hist[-1]
            wins = [gwindow() for _ in range(max(1, args.windows))]
            dtw = sorted(wins)[len(wins) // 2]
            imgs = args.imgs_per_gpu * args.steps
            gstep.sync_optimizer_state()
            print(json.dumps({'img_s': round(imgs / dtw, 3), 'ms_per_step': round(dtw / args.steps * 1e3, 3),
                              'windows_img_s': [round(imgs / w, 1) for w in wins],
                              'steady': bool(max(sorted(wins)[1:-1] or wins) / min(sorted(wins)[1:-1] or wins) - 1.0 < 0.03),
                              'steps_replayed': gstep.steps, 'preheat_s': round(spent, 2),
                              'loss_finite': bool(torch.isfinite(gstep.out['loss']).item())}), flush=True)
            return
    else:
        model.eval()
        if args.config == 'kgdet':
            synthetic.calibrate_scores(model, batch, cfg.test_cfg.score_thr, 0.02, autocast)   # so that decode + NMS have work
        else:
            synthetic.calibrate_scores_serial(model, batch, cfg.test_cfg.score_thr, 0.002, autocast)
        n_det = [0]

        use_graph = bool(args.graph)   # (config 5 too since round 4: the batched device soft-NMS reads nothing back)
        if use_graph:
            # backbone -> head -> decode -> fused NMS as one hipGraph launch + one device->host copy per batch
            run = model.graphed_test_batch(batch['img'], batch['img_meta'], rescale=True,
                                           autocast_dtype=torch.bfloat16 if args.dtype == 'bf16' else None)

        if use_graph:
            run.static_img.copy_(batch['img'])      # the batch is resident in the graph's input buffer (a loader's H2D target)

        # --pipeline 1 (default with the graph): batch k's result copy (2.8 MB device -> page-locked host) and its host-side
        # unpacking run UNDER batch k + 1's kernels (run.submit / run.collect, kgdet_amd/detector.py) -- every batch is still
        # replayed, copied and unpacked inside the timed window (`flush` collects the last one before the closing synchronize)
        pipelined = use_graph and bool(args.pipeline)
        pending = [None]

        def step():
            if pipelined:
                slot = run.submit()
                res = run.collect(pending[0]) if pending[0] is not None else None
                pending[0] = slot
                if res is None:
                    return None
            elif use_graph:
                res = run(run.static_img)
            else:
                with torch.no_grad(), autocast:
                    res = model.simple_test_batch(batch['img'], batch['img_meta'], rescale=True)
            n_det[0] = sum(sum(len(d) for d in r[0]) for r in res)
            return res

        def flush():
            if pending[0] is not None:
                res = run.collect(pending[0])
                pending[0] = None
                n_det[0] = sum(sum(len(d) for d in r[0]) for r in res)

    # Inference runs as a serving loop would: ONE autocast scope around all batches, so autocast's weight cache keeps
    # the bf16 copies of the FPN / head convolution weights instead of re-casting them every batch (44 launches).
    import contextlib
    scope = torch.autocast('cuda', dtype=torch.bfloat16, enabled=args.dtype == 'bf16') if args.mode == 'infer' \
        else contextlib.nullcontext()
    def timed_window():
        """EXACTLY --steps steps between barrier + synchronize on both sides; MAX over ranks"""
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        t0 = time.time()
        for _ in range(args.steps):
            step()
        if args.mode == 'infer':
            flush()          # (pipelined inference: the last batch's results are collected inside the window)
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        dt = time.time() - t0
        if dist_on:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    board = BoardSampler(local_rank) if rank == 0 else None
    preheat = {'windows_img_s': [], 'seconds': 0.0}
    burst = None
    with scope:
        for _ in range(args.warmup):
            step()
        # the objects built so far (model, optimizer state, caches) move to the permanent generation: the collector no longer
        # walks them during the timed windows -- a full collection inside a 0.25 s window of this host-paced loop reads as a
        # 2-3 % dip of that window.  Garbage made by the steps themselves is still collected.
        import gc
        gc.collect()
        gc.freeze()
        if args.mode == 'train' and args.preheat_s > 0:
            # Pre-heat by wall time.  Every decision below is taken on MAX-reduced window times, i.e. on numbers that are
            # identical on all ranks, so all ranks run the same number of steps (the step contains collectives).
            burst = timed_window()                      # the cold board: what round 3 reported as the headline
            hist, spent = [burst], burst
            while spent < args.preheat_max_s:
                last3 = hist[-3:]
                settled = len(last3) == 3 and max(last3) / min(last3) - 1.0 <= args.steady_tol
                if spent >= args.preheat_s and settled:
                    break
                hist.append(timed_window())
                spent += hist[-1]
            preheat = {'windows_img_s': [round(args.imgs_per_gpu * world * args.steps / w, 1) for w in hist],
                       'seconds': round(spent, 2)}
        if board is not None:
            board.start()
        windows = [timed_window() for _ in range(max(1, args.windows))]
        if board is not None:
            board.stop()
        mixed_window = None
        if args.mode == 'train' and not args.mixed_shapes and not dist_on:
            # the step real DeepFashion2 batches take (imgs_per_gpu = 2 mixes aspect ratios): one window on a mixed-shape batch
            keep = batch
            batch = synthetic.make_batch(args.imgs_per_gpu, device, seed=rank, mixed_shapes=True)
            for _ in range(5):
                step()
            mixed_window = sorted(timed_window() for _ in range(3))[1]      # (median of three: one window caught a 28 % hiccup once)
            batch = keep
        exposed, reducer_stats = None, None
        if dist_on and args.mode == 'train':
            red = hook._reducer
            if red is not None and red.buckets:
                before = red.launched_from_hooks
                for _ in range(2):
                    step()
                reducer_stats = (len(red.buckets), (red.launched_from_hooks - before) / 2.0)
            # what the exchange costs the step although it runs under backward: windows with the exchange switched off
            # (local gradients only) ALTERNATING with windows with it, taken after the pre-heat above (steady clock);
            # eight pairs, the median of the PAIRED differences.  AFTER the measurement: the weights diverge between
            # ranks from here on.
            diffs = []
            for _ in range(8):
                hook.set_local_only(True)
                step()
                tl = timed_window()
                hook.set_local_only(False)
                step()
                diffs.append(timed_window() - tl)
            diffs.sort()
            exposed = 0.5 * (diffs[3] + diffs[4]) / args.steps
            exposed_spread = (diffs[-1] - diffs[0]) / args.steps
            bucket_times = None
            if red is not None and red.buckets:      # one traced step: when every bucket went out / came back relative to backward's end
                red._trace_ev, red.trace = [], True
                step()
                red.trace = False
                bucket_times = red.bucket_trace()
    dt = sorted(windows)[len(windows) // 2]

    ar = allreduce_busbw(device, world) if (dist_on and args.mode == 'train') else None
    if ar is not None:
        ar['exposed_ms'] = round(exposed * 1e3, 3)
        ar['exposed_spread_ms'] = round(exposed_spread * 1e3, 3)
        ar['exposed_note'] = ('median over eight alternating window pairs (after the pre-heat) of: step time with the '
                              'overlapped exchange minus the same step without it; spread = max - min of the pairs (at one '
                              'rank the collective is a no-op: this is the cost of the bucket copies and the side stream)')
        ar['bucket_times'] = bucket_times
        ar['buckets'] = reducer_stats[0] if reducer_stats else None
        ar['buckets_issued_inside_backward_per_step'] = reducer_stats[1] if reducer_stats else None
    if rank == 0:
        imgs = args.imgs_per_gpu * world * args.steps
        out = {
            **({'rehearsal': True} if os.environ.get('KGDET_BENCH_REHEARSAL') == '1' else {}),
            'metric': ('training images/sec (whole node) %s R50-FPN 800x1333' if args.mode == 'train'
                       else 'inference images/sec %s R50-FPN 800x1333') % (
                           'KGDet' if args.config == 'kgdet' else 'RepPoints-kp serial (config 5)'),
            'value': round(imgs / dt, 3), 'unit': 'img/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 2), 'higher_is_better': True,
            'windows': {'n': len(windows), 'steps_each': args.steps, 'statistic': 'median',
                        'img_s': [round(imgs / w, 1) for w in windows],
                        'min': round(imgs / max(windows), 1), 'max': round(imgs / min(windows), 1)},
            # steady: the windows agree within 3 % -- judged on the windows WITHOUT the slowest and the fastest one when
            # there are at least five (this loop is paced by the host within a few percent of the GPU time, and the GPU box's
            # host is shared: a neighbour's burst reads as one 0.25 s window at -10 %; all windows are listed above)
            'steady': bool((lambda w: max(w) / min(w) - 1.0 < 0.03)(sorted(windows)[1:-1] if len(windows) >= 5 else windows)),
            'steady_rule': 'max / min - 1 < 3 % over the windows without the slowest and the fastest',
            'burst_img_s': round(imgs / burst, 1) if burst else None,
            'preheat': preheat,
            'board': board.summary() if board is not None else None,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32 (3 MFMA products per multiply on hi/lo-split operands -- fp16 parts forward, bf16 parts for gradients --, f32 accumulate)' if args.dtype == 'fp32'
            else 'bf16 (dense convs and deformable operands, f32 accumulate)',
            'data': 'synthetic',
            **({'numa_node_of_rank0': NUMA_NODE} if NUMA_NODE is not None else {}),
            'config': {'workload': '%s R50-FPN %s step, %d img/GPU at 800x1333 (padded 800x1344), '
                                   '%s' % ('KGDet' if args.config == 'kgdet' else
                                           'RepPoints-kp serial head (config 5, 5-level FPN, soft-NMS)',
                                           args.mode, args.imgs_per_gpu,
                                           ('DeformConv fwd/bwd + focal/moment losses + grad-clip + Adam' +
                                            (' + RCCL grad all-reduce over %d ranks (overlapped with backward)' % world
                                             if dist_on else '; no collective at dp1 (one rank: nothing to exchange)'))
                                           if args.mode == 'train' else
                                           'backbone + FPN + DeformConv head forward + keypoint-guided decode + NMS'),
                       'global_batch': args.imgs_per_gpu * world, 'parallelism': 'dp%d' % world},
        }
        if args.mode == 'infer' and use_graph:
            # the same batch with the host->device copy of the fp32 batch INSIDE the step (mmdetection's loader hands over host
            # tensors): (a) copy, then compute, on one stream; (b) the serving arrangement -- batch k + 1 copied on a side stream
            # into a second device buffer while batch k computes, a device-to-device copy into the graph's input at step start
            host = batch['img'].cpu().pin_memory()
            with scope:
                def serial():
                    run.static_img.copy_(host, non_blocking=True)
                    return run(run.static_img)
                stage = [torch.empty_like(batch['img']) for _ in range(2)]
                side = torch.cuda.Stream()
                ready = [torch.cuda.Event(), torch.cuda.Event()]
                freed = [torch.cuda.Event(), torch.cuda.Event()]
                k = [0]

                def prefetch(slot):
                    with torch.cuda.stream(side):
                        side.wait_event(freed[slot])
                        stage[slot].copy_(host, non_blocking=True)
                        ready[slot].record(side)

                for e in freed:
                    e.record()
                prefetch(0)

                def overlapped():
                    slot = k[0] & 1
                    prefetch(slot ^ 1)
                    torch.cuda.current_stream().wait_event(ready[slot])
                    res = run(stage[slot])        # (D2D copy into the graph's input buffer + replay + results to the host)
                    freed[slot].record()
                    k[0] += 1
                    return res

                def rate(fn):
                    for _ in range(5):
                        fn()
                    torch.cuda.synchronize()
                    t0 = time.time()
                    for _ in range(args.steps):
                        fn()
                    torch.cuda.synchronize()
                    return args.imgs_per_gpu * args.steps / (time.time() - t0)
                pend = [None]

                def overlapped_both():      # ... and batch k's result copy + unpacking under batch k + 1 (run.submit / collect)
                    slot = k[0] & 1
                    prefetch(slot ^ 1)
                    torch.cuda.current_stream().wait_event(ready[slot])
                    out_slot = run.submit(stage[slot])
                    freed[slot].record()
                    k[0] += 1
                    res = run.collect(pend[0]) if pend[0] is not None else None
                    pend[0] = out_slot
                    return res
                flush()
                one_at_a_time = rate(lambda: run(run.static_img))
                both = rate(overlapped_both)
                run.collect(pend[0])
                out['batch_at_a_time'] = {'img_s': round(one_at_a_time, 1),
                                          'note': 'the resident batch replayed, copied to the host and unpacked BEFORE the next '
                                                  'one starts (`value` overlaps batch k\'s result copy and unpacking with batch '
                                                  'k + 1 when config.pipelined is true)'}
                serial_rate = rate(serial)
                out['reference_protocol_img_s'] = round(serial_rate, 1)      # = with_h2d.serial_img_s
                out['reference_protocol_note'] = ('the figure comparable to the reference\'s tools/benchmark.py:84-108 / MODEL_ZOO.md:31 protocol '
                                                  '(data hand-over inside, one batch at a time): the page-locked fp32 batch copied host->device, '
                                                  'then the batch, results on the host before the next one starts')
                out['with_h2d'] = {'serial_img_s': round(serial_rate, 1), 'overlapped_img_s': round(rate(overlapped), 1),
                                   'overlapped_in_and_out_img_s': round(both, 1),
                                   'batch_MB': round(host.numel() * 4 / 1e6, 1),
                                   'note': '`value` times a batch already resident in the graph\'s input buffer; serial = page-locked '
                                           'fp32 batch copied host->device on the compute stream, then the batch; overlapped = the copy of '
                                           'batch k + 1 on a side stream under batch k (two device buffers); overlapped_in_and_out = '
                                           'that, and batch k\'s result copy + host unpacking under batch k + 1'}
        if args.mode == 'infer':
            out['config']['detections_per_image'] = round(n_det[0] / args.imgs_per_gpu, 1)
            out['config']['input'] = ('fp32 batch resident in the captured graph\'s input buffer (run.static_img); results '
                                      'copied to page-locked host memory and unpacked per image and class inside the timed window'
                                      if use_graph else 'fp32 batch resident in HBM')
            out['config']['pipelined'] = bool(pipelined)
        if args.mode == 'train':
            # the whole step against the split-operand MFMA roof: algorithmic flops of one step (tools/step_flops.py: dense + deformable
            # convolutions, forward + the gradients a step computes; frozen stem / layer 1 forward only) over the measured step time
            # (kgdet_step_flops above)
            sf = STEP_FLOPS.get((args.config, args.imgs_per_gpu))
            if sf:
                out['roofline_step'] = {'algorithmic_flops': sf, 'achieved': round(sf * world / (dt / args.steps) / 1e12 / world, 1),
                                        'peak': round(BF16_MFMA_PEAK_TFLOPS / 3.0, 1), 'unit': 'TFLOP/s per GPU',
                                        'frac': round(sf / (dt / args.steps) / 1e12 / (BF16_MFMA_PEAK_TFLOPS / 3.0), 4),
                                        'source': 'bench.py::kgdet_step_flops', 'flops_by_family': kgdet_step_flops(args.imgs_per_gpu)[1] if args.config == 'kgdet' else None,
                                        'kernel_us_by_family': step_families()}
            err = value_error_vs_f64()
            if err is not None and args.config == 'kgdet' and args.dtype == 'fp32':
                out['value_error_vs_f64'] = err
        if args.mode == 'train' and mixed_window is not None:
            out['mixed_shapes'] = {'img_s': round(imgs / mixed_window, 3), 'ms_per_step': round(mixed_window / args.steps * 1e3, 2),
                                   'vs_value': round((imgs / mixed_window) / (imgs / dt), 4),
                                   'note': 'three more timed windows (median) of the same step on a batch whose second image has a smaller '
                                           'pad_shape of its own (synthetic.mixed_shapes_of): invalid grid points, same sync-free path '
                                           '(valid extents in the dense targets / fused loss kernels; tests/test_gpu_head.py)'}
        if not args.no_roofline:
            if args.mode == 'train':
                out['roofline'] = dcn_roofline(device, 2, 'split')
            else:      # the grouped forward the inference batch actually runs
                out['roofline'] = dcn_roofline(device, args.imgs_per_gpu, 'bf16' if args.dtype == 'bf16' else 'split')
        if not args.no_roofline and args.mode == 'train' and args.config == 'kgdet':
            bw = dcn_backward_products_live(device)
            bw_t = dcn_backward_products_live(device, regime='trained')
            bw_s = dcn_backward_products_live(device, regime='step') if os.path.isfile(os.path.join(ROOT, STEP_OFFSETS)) else {}
            rl = out['roofline']
            rl['backward'] = bw
            rl['backward_trained_offsets'] = bw_t
            rl['backward_step_offsets'] = bw_s
            # (scalar copies: a parser that drops nested objects still sees the fractions)
            for name in ('grad_input', 'grad_offset', 'grad_weight'):
                if name in bw:
                    rl['bwd_%s_frac' % name], rl['bwd_%s_us' % name] = bw[name]['frac'], bw[name]['us']
                if name in bw_t:
                    rl['bwd_%s_frac_trained' % name], rl['bwd_%s_us_trained' % name] = bw_t[name]['frac'], bw_t[name]['us']
                if name in bw_s:
                    rl['bwd_%s_frac_step' % name], rl['bwd_%s_us_step' % name] = bw_s[name]['frac'], bw_s[name]['us']
            # the forward as the training step runs it: every call re-packs the six weights (they changed); the inference-style
            # launch above reuses the packed images (`launch_us`: pack NOT inside)
            rl['launch_us_pack_inside'] = bw['forward_with_pack_us']
            rl['frac_pack_inside'] = round(rl['algorithmic_flops'] / (bw['forward_with_pack_us'] * 1e-6) / 1e12 / rl['peak'], 4)
            rl['launch_us_note'] = ('launch_us / frac: dcn_build_taps + dcn_fwd_plane<2> + dcn_fwd_fixup_static with the weight images '
                                    'packed beforehand (an unchanged weight is packed once); *_pack_inside: the same plus '
                                    'dcn_pack_weight_all_multi per call, as in a training step')
        if args.mode == 'train' and args.config == 'kgdet' and not args.no_exact_leg and world == 1 and not dist_on:
            # (one rank only: this block runs on rank 0 alone, a step of an N-rank job contains collectives)
            # the reference's precision class, measured: the same step in plain fp32 arithmetic (f32-input MFMA deformable kernels,
            # MIOpen fp32 convolutions with its heuristic picks -- no find), a short window after the steady ones
            try:
                from kgdet_amd import dcn as _dcn
                torch.backends.cudnn.benchmark = False
                with _dcn.arithmetic('exact'):
                    for _ in range(3):
                        step()
                    torch.cuda.synchronize()
                    t0 = time.time()
                    n_ex = max(5, min(args.steps, 20))
                    for _ in range(n_ex):
                        step()
                    torch.cuda.synchronize()
                    tex = (time.time() - t0) / n_ex
                torch.backends.cudnn.benchmark = bool(args.miopen_find)
                out['exact_fp32'] = {'img_s': round(args.imgs_per_gpu * world / tex, 2), 'ms_per_step': round(tex * 1e3, 2),
                                     'steps': n_ex,
                                     'note': 'the same step under dcn.arithmetic(\'exact\'): exact-fp32 deformable kernels '
                                             '(v_mfma_f32_32x32x2_f32) and MIOpen fp32 dense convolutions (heuristic solver picks); '
                                             '`value` is the split-operand arithmetic named in `dtype`, pinned to a float64 golden of '
                                             'the step (tests/test_gpu_head.py)',
                                     'speedup_of_value': round((imgs / dt) / (args.imgs_per_gpu * world / tex), 3)}
            except Exception as e:
                out['exact_fp32'] = {'error': '%s: %s' % (type(e).__name__, e)}
        if args.mode == 'train' and world == 1 and not dist_on and args.dtype == 'fp32' and args.graph_train:
            gs = graphed_step_leg(args)
            out['graphed_step'] = gs
            out['eager_step'] = {'img_s': out['value'], 'ms_per_step': out['ms_per_step']}
            # `value` is ALWAYS the median of the timed eager windows above (the ones the driver's clock brackets); the replayed
            # graph -- the same kernels in the same order, measured in a child process -- is reported beside it
            out['step_mode'] = 'eager'
        if (args.mode == 'train' and world == 1 and args.config == 'kgdet' and not args.no_inference_leg):
            out['inference'] = inference_leg()
        if ar is not None:
            out['allreduce'] = ar
        if world == 1 and not args.no_cpu_baseline:
            rl = out.get('roofline')
            out['cpu_baseline'] = cpu_baseline(device, rl['launch_us'] * 1e-6 if rl and rl.get('launch_us') and
                                               args.mode == 'train' else None)
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
