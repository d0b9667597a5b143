"""The overlapped gradient reducer on the GPU under RCCL (a one-rank `nccl` group: the only group a 1-GPU box can
form), driving the real KGDet detector: bucket hooks, side HIP stream, multi-tensor pack, RCCL all-reduce,
bucket-view gradients.  Reference: mmdet/core/utils/dist_utils.py:9-58 (flat all-reduce after backward)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from kgdet_amd import configs, synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def nccl_group():
    assert torch.cuda.is_available()
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1)
    yield
    dist.destroy_process_group()


def _detector_and_batch(seed=0):
    from kgdet_amd.registry import build_detector
    cfg = configs.kgdet_r50_fpn()
    torch.manual_seed(seed)
    model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda()
    model.train()
    batch = synthetic.make_batch(2, 'cuda', seed=0, img_shape=(384, 480, 3), pad_shape=(384, 480, 3))
    for k in ('gt_bboxes', 'gt_keypoints'):
        batch[k] = [t.clamp(max=370) for t in batch[k]]
    return cfg, model, batch


def _loss(model, batch):
    losses = model(batch['img'], batch['img_meta'], return_loss=True, gt_bboxes=batch['gt_bboxes'],
                   gt_labels=batch['gt_labels'], gt_keypoints=batch['gt_keypoints'])
    return sum(sum(v) if isinstance(v, (list, tuple)) else v for v in losses.values())


def test_overlapped_reducer_under_rccl_matches_local_gradients_bit_for_bit(nccl_group):
    """With one rank the averaged gradient IS the local gradient, so after finish() every p.grad must equal the
    gradient autograd produced, bit for bit, on every step -- through the hook / side-stream / RCCL path."""
    from kgdet_amd.dist import OverlappedGradReducer
    cfg, model, batch = _detector_and_batch()
    params = [p for p in model.parameters() if p.requires_grad]
    red = OverlappedGradReducer(params, bucket_size_mb=32)
    for step in range(4):
        for p in params:
            p.grad = None
        _loss(model, batch).backward()
        local = [None if p.grad is None else p.grad.clone() for p in params]
        red.finish()
        torch.cuda.synchronize()
        for p, g in zip(params, local):
            assert (p.grad is None) == (g is None)
            if g is not None:
                assert torch.equal(p.grad, g)
        if step >= 1:
            for plist, views in zip(red.buckets, red._views):
                for p, v in zip(plist, views):
                    assert p.grad.data_ptr() == v.data_ptr()        # no copy back: the gradient is the bucket view
    n = sum(p.numel() for b in red.buckets for p in b)
    assert n == 52250071                                             # the 209.0 MB payload of DESIGN.md section 7
    assert len(red.buckets) >= 6
    # all but (possibly) the first-finished bucket started from a hook inside backward, i.e. overlapped
    assert red.launched_from_hooks >= 3 * (len(red.buckets) - 1)


def test_dist_optimizer_hook_overlap_equals_flat_allreduce_over_steps(nccl_group):
    """DistOptimizerHook(overlap=True) (buckets + RCCL on a side stream) against the reference-style flat
    all-reduce after backward, both forced through the one-rank nccl group: the same weights after 3 steps.
    Compared against the run-to-run spread of the flat path itself (MIOpen's backward kernels are not all
    deterministic), and bit-for-bit when that spread is zero."""
    from kgdet_amd.dist import DistOptimizerHook

    def run(overlap):
        cfg, model, batch = _detector_and_batch(seed=0)
        opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4, fused=True)
        hook = DistOptimizerHook(grad_clip=dict(cfg.optimizer_config.grad_clip), overlap=overlap,
                                 bucket_size_mb=32, force_distributed=True)
        for _ in range(3):
            hook.step(model, opt, _loss(model, batch))
        torch.cuda.synchronize()
        if overlap:
            assert hook._reducer is not None and hook._reducer.launched_from_hooks > 0
        return [p.detach().clone() for p in model.parameters()]

    a, a2, b = run(False), run(False), run(True)
    spread = max(float((x - y).abs().max()) for x, y in zip(a, a2))
    diff = max(float((x - y).abs().max()) for x, y in zip(a, b))
    if spread == 0.0:
        assert diff == 0.0
    else:
        # Adam turns a gradient whose sign is decided by summation-order noise into a step of +-lr, so the LARGEST difference
        # over 40 M weights is a heavy-tailed statistic (5x between two draws happens); the root-mean-square is not
        def rms(u, v):
            return (sum(float((x - y).double().square().sum()) for x, y in zip(u, v)) / sum(x.numel() for x in u)) ** 0.5
        # ... and the spread of ONE pair of runs is itself anywhere between 2e-9 and 2e-7 (whether a few signs flipped or
        # not): a wrong exchange would move every weight by ~lr = 1e-4 per step, three orders of magnitude above this floor
        assert rms(a, b) <= max(3 * rms(a, a2), 5e-7), (rms(a, b), rms(a, a2))
        assert diff <= max(20 * spread, 6.1e-4), (diff, spread)


def test_training_step_with_reducer_has_no_host_syncs(nccl_group):
    """The distributed step (hooks + RCCL + clip + fused Adam) must not read back to the host either."""
    from kgdet_amd.dist import DistOptimizerHook
    cfg, model, batch = _detector_and_batch()
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-6, fused=True)
    hook = DistOptimizerHook(grad_clip=dict(cfg.optimizer_config.grad_clip), overlap=True, bucket_size_mb=32,
                             force_distributed=True)
    for _ in range(3):
        hook.step(model, opt, _loss(model, batch))
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode('error')
    try:
        hook.step(model, opt, _loss(model, batch))
    finally:
        torch.cuda.set_sync_debug_mode('default')
    torch.cuda.synchronize()


@pytest.mark.parametrize('max_norm', [35.0, 0.05])
def test_fused_clip_adam_equals_torch_clip_and_adam(max_norm):
    """DistOptimizerHook on csrc/optim.hip (multi-tensor gradient norm + clip + Adam: two passes) against clip_grad_norm_ +
    torch.optim.Adam.step(): parameters, gradients (scaled in place when the clip is active: max_norm 0.05), Adam state and
    step counters after four steps -- equal to rounding; a parameter that never gets a gradient stays untouched"""
    from kgdet_amd import optim
    from kgdet_amd.dist import DistOptimizerHook

    def run(fused):
        torch.manual_seed(0)
        # (Linear layers: their backward is deterministic, unlike MIOpen's convolution gradients; sizes that are not multiples
        #  of the kernels' 4096-element blocks, one tensor larger than a block, biases smaller than a vector)
        net = torch.nn.Sequential(torch.nn.Linear(67, 129), torch.nn.ReLU(), torch.nn.Linear(129, 64), torch.nn.ReLU(),
                                  torch.nn.Linear(64, 5)).cuda()
        net.unused = torch.nn.Parameter(torch.randn(11, device='cuda'))
        opt = torch.optim.Adam(net.parameters(), lr=1e-2, weight_decay=0.0, fused=True)
        hook = DistOptimizerHook(grad_clip=dict(max_norm=max_norm, norm_type=2))
        x = torch.randn(40, 67, device='cuda')
        prev = optim.ENABLED
        optim.ENABLED = fused
        try:
            for _ in range(4):
                hook.step(net, opt, net(x).square().mean())
        finally:
            optim.ENABLED = prev
        torch.cuda.synchronize()
        assert (hook._fused._host_step is not None) == fused
        state = [(opt.state[p]['exp_avg'].clone(), opt.state[p]['exp_avg_sq'].clone(), float(opt.state[p]['step']))
                 for p in net.parameters() if p in opt.state and 'exp_avg' in opt.state[p]]
        return ([p.detach().clone() for p in net.parameters()],
                [None if p.grad is None else p.grad.clone() for p in net.parameters()], state)

    pa, ga, sa = run(True)
    pb, gb, sb = run(False)
    for x, y in zip(pa, pb):
        assert (x - y).abs().max().item() <= 2e-6 * y.abs().max().item() + 1e-9
    for x, y in zip(ga, gb):
        assert (x is None) == (y is None)
        if x is not None:
            assert (x - y).abs().max().item() <= 1e-6 * y.abs().max().item() + 1e-12
    assert len(sa) == len(sb) and len(sa) >= 6
    for (m1, v1, t1), (m2, v2, t2) in zip(sa, sb):
        assert t1 == t2 == 4.0
        assert (m1 - m2).abs().max().item() <= 1e-6 * m2.abs().max().item() + 1e-12
        assert (v1 - v2).abs().max().item() <= 1e-6 * v2.abs().max().item() + 1e-15


def test_fused_clip_adam_follows_a_reloaded_optimizer_state():
    """resume: optimizer.load_state_dict replaces the state tensors (other step count, other moments); the fused path must pick
    the new step count up instead of counting on from its own"""
    from kgdet_amd.dist import DistOptimizerHook

    def make():
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(31, 50), torch.nn.ReLU(), torch.nn.Linear(50, 3)).cuda()
        return net, torch.optim.Adam(net.parameters(), lr=1e-2, fused=True)

    x = torch.randn(16, 31, device='cuda')
    net, opt = make()
    hook = DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2))
    for _ in range(5):
        hook.step(net, opt, net(x).square().mean())
    ckpt_model, ckpt_opt = {k: v.clone() for k, v in net.state_dict().items()}, opt.state_dict()
    import copy
    ckpt_opt = copy.deepcopy(ckpt_opt)
    for _ in range(3):
        hook.step(net, opt, net(x).square().mean())            # steps 6-8, then "resume" from the step-5 checkpoint
    want = None
    for same_hook in (True, False):
        net.load_state_dict(ckpt_model)
        opt.load_state_dict(copy.deepcopy(ckpt_opt))
        h = hook if same_hook else DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2))
        for _ in range(2):
            h.step(net, opt, net(x).square().mean())
        got = [p.detach().clone() for p in net.parameters()]
        assert float(opt.state[next(iter(net.parameters()))]['step']) == 7.0
        if want is None:
            want = got
        else:
            for a_, b_ in zip(got, want):
                assert torch.equal(a_, b_)


def test_fused_clip_adam_after_a_step_through_torch():
    """a step that falls back to optimizer.step() (here: the fused path switched off for one step) advances
    torch's step tensors; the fused path must re-read them instead of continuing its own count (bias corrections for
    t - 1 otherwise)."""
    from kgdet_amd import optim
    from kgdet_amd.dist import DistOptimizerHook

    def run(pattern):
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(31, 50), torch.nn.ReLU(), torch.nn.Linear(50, 3)).cuda()
        opt = torch.optim.Adam(net.parameters(), lr=1e-2, fused=True)
        hook = DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2))
        x = torch.randn(16, 31, device='cuda', generator=torch.Generator(device='cuda').manual_seed(1))
        prev = optim.ENABLED
        try:
            for on in pattern:
                optim.ENABLED = on
                hook.step(net, opt, net(x).square().mean())
        finally:
            optim.ENABLED = prev
        return [p.detach().clone() for p in net.parameters()], float(opt.state[next(iter(net.parameters()))]['step'])

    pa, ta = run([True, True, True, False, True, True])
    pb, tb = run([False] * 6)
    assert ta == tb == 6.0
    for x, y in zip(pa, pb):
        assert (x - y).abs().max().item() <= 2e-6 * y.abs().max().item() + 1e-9


def test_fused_clip_adam_bumps_parameter_versions():
    """the kernel writes parameters through raw pointers; caches keyed by `_version` (folded BatchNorm weights,
    DeformConv packs) must see the update"""
    from kgdet_amd.dist import DistOptimizerHook
    torch.manual_seed(0)
    net = torch.nn.Linear(8, 4).cuda()
    opt = torch.optim.Adam(net.parameters(), lr=1e-2, fused=True)
    hook = DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2))
    x = torch.randn(5, 8, device='cuda')
    for _ in range(2):
        hook.step(net, opt, net(x).square().mean())
    assert hook._fused._host_step is not None
    v0 = net.weight._version
    hook.step(net, opt, net(x).square().mean())
    assert net.weight._version > v0


def test_bench_force_dist_runs_the_n_rank_path_on_one_gpu():
    """`python bench.py --gpus 1 --force-dist`: launch_ranks -> child torch.distributed.run -> init_process_group('nccl')
    -> broadcast -> DistOptimizerHook(overlap, force_distributed) -> barriers -> MAX-reduced windows -> allreduce_busbw;
    the JSON line carries the `allreduce` object (round-2 review item 1: the world > 1 branch had never executed)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MIOPEN_USER_DB_PATH')}
    res = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--force-dist', '--steps', '4',
                          '--warmup', '3', '--windows', '2', '--no-cpu-baseline', '--no-roofline', '--no-inference-leg'],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    line = [l for l in res.stdout.decode().splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 1 and d['value'] > 0 and d['scaling'] == 'weak'
    assert d['windows']['n'] == 2 and len(d['windows']['img_s']) == 2
    ar = d['allreduce']
    assert ar['payload_MB'] > 200 and ar['ms'] > 0 and 'exposed_ms' in ar
    assert ar['buckets'] >= 6 and ar['buckets_issued_inside_backward_per_step'] >= ar['buckets'] - 1


@pytest.mark.parametrize('mode', ['train', 'infer'])
def test_bench_two_rank_rehearsal_completes(mode):
    """`KGDET_BENCH_REHEARSAL=1 python bench.py --gpus 2`: the whole TWO-rank flow of the bench -- rank start-up, weight broadcast,
    the reducer's bucket exchanges and its late-joiner bit inside the timed windows, the exposed-time windows, the barriers, the
    legs rank 0 runs alone -- on ONE device over gloo (RCCL refuses two ranks on a device).  Not a measurement (`rehearsal: true`
    on the line): it is what catches a leg that steps a multi-rank job from rank 0 alone (a hang, here a timeout)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MIOPEN_USER_DB_PATH')}
    env['KGDET_BENCH_REHEARSAL'] = '1'
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '2', '--windows', '2',
           '--preheat-s', '0', '--no-roofline', '--no-cpu-baseline']
    if mode == 'infer':
        cmd += ['--mode', 'infer', '--dtype', 'bf16', '--imgs-per-gpu', '2']
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    d = json.loads([l for l in res.stdout.decode().splitlines() if l.startswith('{')][-1])
    assert d['rehearsal'] is True and d['n_gpus'] == 2 and d['value'] > 0 and d['config']['parallelism'] == 'dp2'
    if mode == 'train':
        assert d['allreduce']['buckets'] >= 6 and 'exact_fp32' not in d and 'graphed_step' not in d and 'inference' not in d


def test_bench_two_ranks_over_rccl_when_the_box_has_two_gpus():
    """`python bench.py --gpus 2` over RCCL -- the real thing the driver's SCALE run starts: two processes, two devices,
    `init_process_group('nccl')`, the weight broadcast, the bucketed exchange under backward, MAX-reduced windows.  Skipped on the
    one-GPU boxes of this pool; on an 8-GPU driver box it is the first time RCCL sees a peer, BEFORE the scaling bench does
    (round-5 review, item 7)."""
    import json
    import os
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs (RCCL refuses two ranks on one device)')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MIOPEN_USER_DB_PATH',
                                                            'KGDET_BENCH_REHEARSAL')}
    res = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '3',
                          '--windows', '2', '--preheat-s', '0', '--no-roofline', '--no-cpu-baseline'],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    d = json.loads([l for l in res.stdout.decode().splitlines() if l.startswith('{')][-1])
    assert 'rehearsal' not in d and d['n_gpus'] == 2 and d['value'] > 0 and d['config']['parallelism'] == 'dp2'
    ar = d['allreduce']
    assert ar['payload_MB'] > 200 and ar['ms'] > 0 and ar['buckets'] >= 6
    assert ar['buckets_issued_inside_backward_per_step'] >= ar['buckets'] - 1
    assert 'exact_fp32' not in d and 'graphed_step' not in d and 'inference' not in d      # (legs of a one-rank run)
