"""world_size-2 gloo test of the data-parallel gradient exchange (runs on CPU)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except BaseException:
        import traceback
        traceback.print_exc()
        q.put((rank, False))
        raise


def _worker_body(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from kgdet_amd.dist import DistOptimizerHook, OverlappedGradReducer, allreduce_grads
    torch.manual_seed(0)
    model = nn.Sequential(nn.Linear(8, 16), nn.ReLU(), nn.Linear(16, 4), nn.Linear(4, 4))
    unused = nn.Linear(3, 3)                       # never receives a gradient (like FPN2's dead branches)
    params = list(model.parameters()) + list(unused.parameters())
    x = torch.randn(5, 8, generator=torch.Generator().manual_seed(100 + rank))

    def local_grads():
        for p in params:
            p.grad = None
        model(x).pow(2).sum().backward()
        return [p.grad.clone() for p in model.parameters()]

    # reference-style flat all-reduce
    g_local = local_grads()
    allreduce_grads(params)
    g_ref = [p.grad.clone() for p in model.parameters()]
    gathered = [None] * world
    dist.all_gather_object(gathered, [g.numpy() for g in g_local])
    ok = True
    for i, g in enumerate(g_ref):
        mean = sum(torch.from_numpy(gathered[r][i]) for r in range(world)) / world
        ok &= bool(torch.allclose(g, mean, atol=1e-6))
    ok &= all(p.grad is None for p in unused.parameters())

    # overlapped reducer: step 1 discovers live params, later steps use hooks + buckets
    red = OverlappedGradReducer(params, bucket_size_mb=0.0005)
    for step in range(3):
        local_grads()
        red.finish()
        for p, g in zip(model.parameters(), g_ref):
            ok &= bool(torch.allclose(p.grad, g, atol=1e-6))
    ok &= len(red.buckets) > 1
    ok &= red.launched_from_hooks >= len(red.buckets)      # exchanges started inside backward, not in finish()
    ok &= all(p.grad.data_ptr() == v.data_ptr() for pl, vl in zip(red.buckets, red._views) for p, v in zip(pl, vl))

    # a bucketed parameter without a gradient in one step contributes zeros (used to raise / hang) ...
    for p in params:
        p.grad = None
    model[0](x).pow(2).sum().backward()            # only the first Linear receives gradients
    g0 = [p.grad.clone() for p in model[0].parameters()]
    red.finish()
    dist.all_gather_object(gathered, [g.numpy() for g in g0])
    for i, p in enumerate(model[0].parameters()):
        mean = sum(torch.from_numpy(gathered[r][i]) for r in range(world)) / world
        ok &= bool(torch.allclose(p.grad, mean, atol=1e-6))
    # (zeros in the exchange; on this rank the gradient stays None, as under the reference's filter, so the
    # optimizer does not move the parameter on momentum)
    ok &= all(p.grad is None for p in model[2].parameters())
    dead = {id(p) for p in model[2].parameters()}
    ok &= all(float(v.abs().max()) == 0.0 for pl, vl in zip(red.buckets, red._views) for p, v in zip(pl, vl)
              if id(p) in dead)
    # ... and a parameter that starts to receive gradients later is picked up (buckets rebuild) -- at N ranks one step after it
    # showed up: the step in which it appears drops its gradient on every rank (the decision to rebuild is collective, dist.py (c))
    for p in params:
        p.grad = None
    (model(x).pow(2).sum() + unused(x[:, :3]).pow(2).sum()).backward()
    red.finish()
    ok &= all(p.grad is None for p in unused.parameters()) and red.late_joiner_rebuilds == 0
    for p in params:
        p.grad = None
    (model(x).pow(2).sum() + unused(x[:, :3]).pow(2).sum()).backward()
    gu = [p.grad.clone() for p in unused.parameters()]
    red.finish()
    ok &= red.late_joiner_rebuilds == 1
    dist.all_gather_object(gathered, [g.numpy() for g in gu])
    for i, p in enumerate(unused.parameters()):
        mean = sum(torch.from_numpy(gathered[r][i]) for r in range(world)) / world
        ok &= bool(torch.allclose(p.grad, mean, atol=1e-6))
    ok &= all(p in red._bucket_of for p in unused.parameters())
    # RCCL calls are issued in bucket-index order whatever order autograd finishes the gradients in (a rank whose
    # ready queue breaks a tie differently must not issue a different sequence of collectives)
    order = []
    launch = red._launch
    red._launch = lambda b: (order.append(b), launch(b))[1]
    local_grads()          # the hooks fired in autograd's order ...
    red.finish()
    ok &= order == list(range(len(red.buckets)))
    order.clear()
    for p in params:
        p.grad = None
    with torch.no_grad():
        for p in model.parameters():
            p.grad = torch.ones_like(p) * (rank + 1)
    for p in sorted(red.active, key=lambda p: -red._bucket_of[p]):     # ... and here in exactly the reverse bucket order
        red._on_grad(p)
        ok &= order == sorted(order)
    nb_before_finish = len(order)
    red.finish()
    ok &= order == list(range(len(red.buckets))) and nb_before_finish == len(red.buckets)
    ok &= all(bool(torch.allclose(p.grad, torch.full_like(p, 1.5))) for p in model.parameters())
    red._launch = launch
    # two backward passes before finish() would exchange a partial gradient: refused loudly
    local_grads()
    try:
        model(x).pow(2).sum().backward()
        ok = False
    except RuntimeError as e:
        ok &= 'fired twice' in str(e)
    red.finish()
    # non-contiguous gradients are packed by value, not by storage order
    wt = nn.Parameter(torch.randn(4, 6, generator=torch.Generator().manual_seed(7)))
    red2 = OverlappedGradReducer([wt], bucket_size_mb=1)
    for step in range(2):
        wt.grad = (torch.arange(24.).reshape(6, 4) * (rank + 1)).t()     # strided
        if step:
            red2._on_grad(wt)
        red2.finish()
        ok &= bool(torch.equal(wt.grad, torch.arange(24.).reshape(6, 4).t() * (sum(range(1, world + 1)) / world)))

    # sentinel mode: with begin_step() after the gradients were set to None, ONE hook per bucket is left after a full-hook
    # step; results stay the averaged gradients, launches stay in bucket order and inside backward; a step whose
    # gradients are not None at begin_step, or without begin_step, launches nothing early; a missing gradient delays to finish()
    red.close()
    red3 = OverlappedGradReducer(params, bucket_size_mb=0.0005)
    for step in range(5):
        for p in params:
            p.grad = None
        red3.begin_step()
        model(x).pow(2).sum().backward()
        red3.finish()
        for p, g in zip(model.parameters(), g_ref):
            ok &= bool(torch.allclose(p.grad, g, atol=1e-6))
    ok &= red3._sentinel and len(red3._hooks) == len(red3.buckets)
    order3 = []
    launch3 = red3._launch
    red3._launch = lambda b: (order3.append(b), launch3(b))[1]
    for p in params:
        p.grad = None
    red3.begin_step()
    model(x).pow(2).sum().backward()
    ok &= order3 == list(range(len(red3.buckets)))          # all out before finish(), in index order
    red3.finish()
    for p, g in zip(model.parameters(), g_ref):
        ok &= bool(torch.allclose(p.grad, g, atol=1e-6))
    order3.clear()
    for p in params:                                         # no begin_step: nothing may leave from the hooks
        p.grad = None
    model(x).pow(2).sum().backward()
    ok &= order3 == []
    red3.finish()
    ok &= order3 == list(range(len(red3.buckets)))
    for p, g in zip(model.parameters(), g_ref):
        ok &= bool(torch.allclose(p.grad, g, atol=1e-6))
    order3.clear()
    red3.begin_step()                                        # gradients still set: begin_step must refuse to vouch
    ok &= not red3._began
    for p in params:
        p.grad = None
    red3.begin_step()
    model[0](x).pow(2).sum().backward()                      # later layers get no gradient: their buckets wait for finish()
    n_early = len(order3)
    red3.finish()
    ok &= order3 == list(range(len(red3.buckets))) and n_early < len(red3.buckets)
    red3._launch = launch3
    red3.close()

    # the optimizer hook end to end: both ranks stay in sync
    opt = torch.optim.SGD(params, lr=0.1)
    hook = DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2), overlap=True, bucket_size_mb=1)
    for _ in range(4):
        hook.step(model, opt, model(x).pow(2).sum())
    ok &= hook._reducer._sentinel
    w = [p.detach().clone() for p in model.parameters()]
    dist.all_gather_object(gathered, [t.numpy() for t in w])
    for i in range(len(w)):
        ok &= bool(torch.allclose(torch.from_numpy(gathered[0][i]), torch.from_numpy(gathered[1][i]), atol=1e-6))
    q.put((rank, ok))
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == {0: True, 1: True}


def _worker_enforce(rank, world, port, q):
    """Round-4 hardening (ADVICE r3 dist.py:251): the preconditions of the overlapped exchange are
    enforced instead of documented."""
    ok = True
    try:
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        dist.init_process_group('gloo', rank=rank, world_size=world)
        from kgdet_amd.dist import OverlappedGradReducer
        torch.manual_seed(0)
        model = nn.Sequential(nn.Linear(8, 16), nn.ReLU(), nn.Linear(16, 4))
        extra = nn.Linear(8, 2)
        params = list(model.parameters()) + list(extra.parameters())
        x = torch.randn(5, 8, generator=torch.Generator().manual_seed(100 + rank))

        # (a) ranks that would cut different buckets (rank 1 also trains `extra`) fail loudly at build time, on BOTH ranks,
        #     instead of hanging in the first collective whose sizes differ
        loss = model(x).pow(2).sum() + (extra(x).pow(2).sum() if rank == 1 else 0.0)
        loss.backward()
        red = OverlappedGradReducer(params, bucket_size_mb=0.0005)
        try:
            red.finish()
            ok = False
        except RuntimeError as e:
            ok &= 'disagree on the gradient buckets' in str(e)
        red.close()

        # (b) same buckets everywhere; then one rank misses one parameter's gradient in a step: the exchange itself goes
        #     through (zeros), the NEXT finish() raises on every rank
        for p in params:
            p.grad = None
        red = OverlappedGradReducer(list(model.parameters()), bucket_size_mb=0.0005)
        for _ in range(2):
            for p in model.parameters():
                p.grad = None
            model(x).pow(2).sum().backward()
            red.finish()
        ok &= red.layout_digest[0] == len(red.buckets)
        for p in model.parameters():
            p.grad = None

        def without_bias():                          # the last layer's bias takes no part: its gradient stays None
            return torch.nn.functional.linear(torch.relu(model[0](x)), model[2].weight).pow(2).sum()
        (without_bias() if rank == 1 else model(x).pow(2).sum()).backward()     # lost on rank 1 only
        red.finish()                                 # completes: zeros from rank 1
        for p in model.parameters():
            p.grad = None
        model(x).pow(2).sum().backward()
        try:
            red.finish()
            ok = False
        except RuntimeError as e:
            ok &= 'some ranks only' in str(e)
        # a parameter without a gradient on EVERY rank is legal (count 0): no error
        red.close()
        red = OverlappedGradReducer(list(model.parameters()), bucket_size_mb=0.0005)
        for step in range(4):
            for p in model.parameters():
                p.grad = None
            (without_bias() if step == 2 else model(x).pow(2).sum()).backward()
            red.finish()
        q.put((rank, bool(ok)))
        dist.destroy_process_group()
    except BaseException:
        import traceback
        traceback.print_exc()
        q.put((rank, False))
        raise


def test_two_rank_preconditions_are_enforced():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_enforce, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == {0: True, 1: True}


def _worker_late(rank, world, port, q):
    """Round 5 (dist.py): a parameter outside the buckets that starts to receive a gradient -- the rebuild is a collective
    decision, one step late, and a late joiner on ONE rank only raises on both instead of mismatching the collectives."""
    ok = True
    try:
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        dist.init_process_group('gloo', rank=rank, world_size=world)
        from kgdet_amd.dist import OverlappedGradReducer
        torch.manual_seed(0)
        model = nn.Sequential(nn.Linear(8, 16), nn.ReLU(), nn.Linear(16, 4))
        extra = nn.Linear(8, 2)                       # joins at step 2
        params = list(model.parameters()) + list(extra.parameters())
        x = torch.randn(5, 8, generator=torch.Generator().manual_seed(100 + rank))

        def step(red, with_extra):
            for p in params:
                p.grad = None
            loss = model(x).pow(2).sum()
            if with_extra:
                loss = loss + extra(x).pow(2).sum()
            loss.backward()
            local = [None if p.grad is None else p.grad.clone() for p in extra.parameters()]
            red.finish()
            return local

        # (1) the branch becomes active on BOTH ranks in step 2: dropped in step 2 (no rank updates it), every rank rebuilds
        #     in step 3, from then on its gradient is the rank average
        red = OverlappedGradReducer(params, bucket_size_mb=0.0005)
        for s in range(5):
            local = step(red, with_extra=s >= 2)
            if s < 2:
                ok &= all(p.grad is None for p in extra.parameters())
            elif s == 2:
                ok &= all(p.grad is None for p in extra.parameters()) and red.late_joiner_rebuilds == 0
            else:
                ok &= red.late_joiner_rebuilds == 1
                gathered = [None] * world
                dist.all_gather_object(gathered, [g.numpy() for g in local])
                for i, p in enumerate(extra.parameters()):
                    mean = sum(torch.from_numpy(gathered[r][i]) for r in range(world)) / world
                    ok &= p.grad is not None and bool(torch.allclose(p.grad, mean, atol=1e-6))
        n_before = len(red.buckets)
        red.close()

        # (2) the branch becomes active on rank 1 ONLY: step 2 completes on both ranks (rank 1 drops the gradient), step 3
        #     raises on BOTH ranks (the ranks would cut different buckets) -- no hang, no mismatched collective
        red = OverlappedGradReducer(params, bucket_size_mb=0.0005)
        for s in range(2):
            step(red, with_extra=False)
        step(red, with_extra=(rank == 1))
        ok &= all(p.grad is None for p in extra.parameters())
        try:
            step(red, with_extra=(rank == 1))
            ok = False
        except RuntimeError as e:
            ok &= 'disagree on the gradient buckets' in str(e)
        ok &= n_before >= 1
        q.put((rank, bool(ok)))
        dist.destroy_process_group()
    except BaseException:
        import traceback
        traceback.print_exc()
        q.put((rank, False))
        raise


def test_two_rank_late_joiner_is_a_collective_decision():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_late, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == {0: True, 1: True}
