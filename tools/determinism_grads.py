"""Which gradients differ between two identical eager forward/backward passes from the same weights?  Gradient of a parameter =
function of everything downstream of it, so the LAST differing modules (in forward order) point at the non-deterministic kernels.
python tools/determinism_grads.py [kgdet|serial]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import build_detector, configs, synthetic
from kgdet_amd.runner import batch_processor
which = sys.argv[1] if len(sys.argv) > 1 else 'kgdet'
batch = synthetic.make_batch(2, 'cuda', seed=0)
cfg = configs.kgdet_r50_fpn() if which == 'kgdet' else configs.reppoints_kp_r50_fpn(soft_nms=True)
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()


def grads():
    model.zero_grad(set_to_none=True)
    out = batch_processor(model, batch)
    out['loss'].backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}, float(out['loss'])


for _ in range(int(os.environ.get('WARM', '1'))):
    grads()     # (first passes: kernel selection, packs, folded-BatchNorm pairs joining the step scope)
a, la = grads()
b, lb = grads()
bad = [(n, float((a[n] - b[n]).abs().max() / (a[n].abs().max() + 1e-30))) for n in a if not torch.equal(a[n], b[n])]
print('loss equal: %s   gradients differing: %d of %d' % (la == lb, len(bad), len(a)))
for n, d in bad:
    print('  %-70s %.2e' % (n, d))

if os.environ.get('HOOKS'):
    # which modules' OUTPUT gradients differ (full backward hooks on the head's and the neck's leaf modules, per call)
    store = {}
    calls = {}

    def hook(name):
        def f(mod, gin, gout):
            k = calls.get(name, 0)
            calls[name] = k + 1
            if gout[0] is not None:
                store.setdefault((name, k), []).append(gout[0].detach().clone())
        return f
    for n, m in list(model.bbox_head.named_modules()) + list(model.neck.named_modules()):
        if len(list(m.children())) == 0:
            m.register_full_backward_hook(hook(n))
    for _ in range(2):
        calls.clear()
        grads()
    for (n, k), v in store.items():
        if len(v) == 2 and not torch.equal(v[0], v[1]):
            print('grad_output differs: %-40s call %d  shape %s  %.2e' % (n, k, tuple(v[0].shape), float((v[0] - v[1]).abs().max() / (v[0].abs().max() + 1e-30))))

if os.environ.get('NODES'):
    # every autograd node's incoming gradient (bit checksum) in execution order, two passes: the first node whose incoming gradient
    # differs was fed by the non-deterministic kernel
    def run_nodes():
        model.zero_grad(set_to_none=True)
        out = batch_processor(model, batch)
        seen, order, stack, log = set(), [], [out['loss'].grad_fn], []
        while stack:
            nd = stack.pop()
            if nd is None or nd in seen:
                continue
            seen.add(nd)
            order.append(nd)
            for nxt, _ in nd.next_functions:
                stack.append(nxt)
        for idx, nd in enumerate(order):
            def pre(gouts, idx=idx, nd=nd):
                for g in gouts:
                    if g is not None and g.is_floating_point():
                        log.append((idx, type(nd).__name__, tuple(g.shape), g.detach().float().contiguous().view(-1).view(torch.int32).to(torch.int64).sum().item()))
                        break
            nd.register_prehook(pre)
        out['loss'].backward()
        torch.cuda.synchronize()
        return log
    l1, l2 = run_nodes(), run_nodes()
    print('nodes executed', len(l1), len(l2))
    shown = 0
    first = next((i for i, (x, y) in enumerate(zip(l1, l2)) if x[3] != y[3]), None)
    if first is not None:
        for x in l1[max(0, first - 14):first]:
            print('  (same)  node %5d %-40s %s' % (x[0], x[1], x[2]))
    for x, y in zip(l1, l2):
        if x[:3] != y[:3]:
            print('graphs diverge', x, y); break
        if x[3] != y[3]:
            print('incoming gradient differs: node %5d %-40s %s' % (x[0], x[1], x[2]))
            shown += 1
            if shown >= 12:
                break
