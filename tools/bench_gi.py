"""grad_input plane kernel timing: python tools/bench_gi.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import dcn
dev = torch.device('cuda:0'); torch.manual_seed(0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for (B, k) in [(2, 3), (2, 5), (2, 7), (8, 7)]:
    C, H, W = 256, 25, 42
    x = torch.randn(B, C, H, W, device=dev)
    off = torch.randn(B, 2 * k * k, H, W, device=dev) * 2
    w = torch.randn(C, C, k, k, device=dev) * 0.01
    go = torch.randn(B, C, H, W, device=dev)
    shape = dcn._shape(x, w, (1, 1), (k // 2, k // 2), (1, 1), 1, 1)
    packed = dcn.pack_weight(w, shape)
    for _ in range(3): dcn.grad_input_plane(x.shape, off, None, w, go, shape, packed)
    torch.cuda.synchronize(); e0.record()
    for _ in range(30): dcn.grad_input_plane(x.shape, off, None, w, go, shape, packed)
    e1.record(); torch.cuda.synchronize()
    print('B=%d k=%d grad_input_plane %.1f us' % (B, k, e0.elapsed_time(e1) / 30 * 1e3), flush=True)
for (B, k) in [(2, 3), (2, 5), (2, 7), (8, 7)]:
    C, H, W = 256, 25, 42
    x = torch.randn(B, C, H, W, device=dev)
    off = torch.randn(B, 2 * k * k, H, W, device=dev) * 2
    w = torch.randn(C, C, k, k, device=dev) * 0.01
    go = torch.randn(B, C, H, W, device=dev)
    shape = dcn._shape(x, w, (1, 1), (k // 2, k // 2), (1, 1), 1, 1)
    packed = dcn.pack_weight(w, shape)
    for _ in range(3): dcn.grad_offset_plane(x, off, w, go, shape, packed)
    torch.cuda.synchronize(); e0.record()
    for _ in range(30): dcn.grad_offset_plane(x, off, w, go, shape, packed)
    e1.record(); torch.cuda.synchronize()
    print('B=%d k=%d grad_offset_plane %.1f us' % (B, k, e0.elapsed_time(e1) / 30 * 1e3), flush=True)
