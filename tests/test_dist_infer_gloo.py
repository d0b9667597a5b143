"""world_size-2 gloo tests (CPU) of the multi-rank inference harness -- runner.multi_gpu_test / collect_results, the semantics of
mmdetection/tools/test.py:38-100 -- and of the overlapped gradient exchange in sentinel mode with DIFFERENT hook arrival orders
on the two ranks."""
import os
import socket

import numpy as np
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


def _spawn(target, world=2, timeout=420):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=timeout) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    return res


class _Subset(object):
    def __init__(self, data, indices):
        self.data, self.indices = data, list(indices)

    def __len__(self):
        return len(self.indices)

    def __getitem__(self, i):
        return self.data[self.indices[i]]


def _same_result(a, b):
    if len(a) != len(b):
        return False
    if len(a) == 1:
        return all(np.array_equal(x, y) for x, y in zip(a[0], b[0]))
    return (all(np.array_equal(x, y) for x, y in zip(a[0], b[0])) and np.array_equal(a[1], b[1])
            and all(np.array_equal(x, y) for x, y in zip(a[2], b[2])))


def _worker_infer(rank, world, port, q):
    try:
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        torch.set_num_threads(2)
        dist.init_process_group('gloo', rank=rank, world_size=world)
        from kgdet_amd import runner
        from tests import cpu_ops
        from tests.golden import demo_cases
        ok = True
        # the sharding rule itself: DistributedSampler(shuffle=False) -- padded by wrapping, strided by rank
        ok &= runner.test_shard(5, 2, 0) == [0, 2, 4] and runner.test_shard(5, 2, 1) == [1, 3, 0]
        ok &= runner.test_shard(4, 2, 1) == [1, 3] and runner.test_shard(1, 2, 1) == [0]
        # collect_results with plain objects: dataset order on rank 0, padding dropped, None elsewhere
        part = ['r%d-%d' % (rank, k) for k in range(3)]
        got = runner.collect_results(part, 5)
        ok &= (got == ['r0-0', 'r1-0', 'r0-1', 'r1-1', 'r0-2']) if rank == 0 else (got is None)
        # the demo detector on the CPU (test-side ops) over an ODD number of demo images: rank 1's last sample is padding
        cfg, model = demo_cases.demo_detector()
        data = _Subset(demo_cases.demo_dataset(test_mode=True), [9, 0, 13])
        with cpu_ops.patched():
            both = runner.multi_gpu_test(model, data, rescale=True)
            if rank == 0:
                alone = runner.single_gpu_test(model, data, rescale=True)
                ok &= both is not None and len(both) == 3 == len(alone)
                ok &= all(_same_result(a, b) for a, b in zip(both, alone))
                ok &= sum(len(np.concatenate(r[0])) for r in alone if len(r) == 3) > 0      # (not vacuous: detections exist)
            else:
                ok &= both is None
        q.put((rank, bool(ok)))
        dist.barrier()
        dist.destroy_process_group()
    except BaseException:
        import traceback
        traceback.print_exc()
        q.put((rank, False))
        raise


def test_two_rank_inference_equals_single_process():
    assert _spawn(_worker_infer) == {0: True, 1: True}


def _worker_sentinel(rank, world, port, q):
    try:
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        dist.init_process_group('gloo', rank=rank, world_size=world)
        from kgdet_amd.dist import OverlappedGradReducer
        torch.manual_seed(0)
        # two independent branches: the rank decides which one is evaluated last -- autograd runs the later branch's backward
        # first, so the gradients (and with them the sentinel hooks) arrive in a different order on the two ranks
        a = nn.Sequential(nn.Linear(8, 16), nn.ReLU(), nn.Linear(16, 4))
        b = nn.Sequential(nn.Linear(8, 12), nn.ReLU(), nn.Linear(12, 4))
        params = list(a.parameters()) + list(b.parameters())
        x = torch.randn(5, 8, generator=torch.Generator().manual_seed(100 + rank))

        def loss():
            return (a(x).pow(2).sum() + b(x).pow(2).sum()) if rank == 0 else (b(x).pow(2).sum() + a(x).pow(2).sum())

        for p in params:
            p.grad = None
        loss().backward()
        local = [p.grad.clone() for p in params]
        gathered = [None] * world
        dist.all_gather_object(gathered, [g.numpy() for g in local])
        want = [sum(torch.from_numpy(gathered[r][i]) for r in range(world)) / world for i in range(len(params))]

        red = OverlappedGradReducer(params, bucket_size_mb=0.0005)
        arrivals, launches = [], []
        launch = red._launch
        red._launch = lambda bk: (launches.append(bk), launch(bk))[1]
        for p in params:      # (the test's own hooks: the order in which autograd finishes the gradients on this rank)
            p.register_post_accumulate_grad_hook(lambda p: arrivals.append(red._bucket_of.get(p) if red.buckets else None))
        ok = True
        orders = []
        for step in range(6):            # step 0 learns the live set, step 1 runs on full hooks, sentinel mode from then on
            for p in params:
                p.grad = None
            arrivals.clear(); launches.clear()
            red.begin_step()
            loss().backward()
            red.finish()
            ok &= launches == list(range(len(red.buckets)))            # the same sequence of collectives on both ranks
            for p, g in zip(params, want):
                ok &= bool(torch.allclose(p.grad, g, atol=1e-6))
            if step >= 2:
                orders.append([bk for bk in arrivals if bk is not None])
        ok &= red._sentinel and len(red.buckets) >= 4
        # the arrival orders really differed between the ranks in the sentinel steps (at least three of them)
        dist.all_gather_object(gathered, orders)
        differing = sum(1 for s in range(len(orders)) if gathered[0][s] != gathered[1][s])
        ok &= len(orders) >= 3 and differing >= 3
        red.close()
        if not ok:
            print('rank', rank, 'sentinel', red._sentinel, 'buckets', len(red.buckets), 'orders', orders, 'differing', differing)
        q.put((rank, bool(ok)))
        dist.destroy_process_group()
    except BaseException:
        import traceback
        traceback.print_exc()
        q.put((rank, False))
        raise


def test_sentinel_mode_with_different_arrival_orders_on_the_ranks():
    assert _spawn(_worker_sentinel, timeout=180) == {0: True, 1: True}
