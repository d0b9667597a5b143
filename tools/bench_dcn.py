"""DeformConv micro-benchmark (SURVEY 8d shapes): time, MFMA and HBM roofline fractions."""
import argparse
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kgdet_amd import dcn

FP32_MFMA_PEAK = 157.3e12
HBM_PEAK = 8.0e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--batch', type=int, default=2)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    rows = []
    for (B, C, H, W, k) in [(args.batch, 256, 25, 42, 3), (args.batch, 256, 25, 42, 5),
                            (args.batch, 256, 25, 42, 7), (8, 256, 25, 42, 7), (2, 256, 100, 168, 3)]:
        K = k * k
        x = torch.randn(B, C, H, W, device=dev)
        off = torch.randn(B, 2 * K, H, W, device=dev) * 2
        w = torch.randn(C, C, k, k, device=dev) * 0.01
        shape = dcn._shape(x, w, (1, 1), (k // 2, k // 2), (1, 1), 1, 1)
        packed = dcn.pack_weight(w, shape)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tf = {}
        for prec in ('split', 'bf16', 'exact'):
            with dcn.forward_precision(prec):
                for _ in range(3):
                    dcn._forward(x, off, None, w, None, shape, packed=packed)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(args.iters):
                    dcn._forward(x, off, None, w, None, shape, packed=packed)
                e1.record()
                torch.cuda.synchronize()
            tf[prec] = e0.elapsed_time(e1) / args.iters * 1e-3
        t = tf['split']
        flops = 2.0 * C * C * K * B * H * W
        byts = 4.0 * (2 * B * C * H * W + 2 * B * K * H * W + C * C * K)
        # backward
        go = torch.randn(B, C, H, W, device=dev)
        needs_i = dict(input=True, offset=True, mask=False, weight=False, bias=False)
        needs_w = dict(input=False, offset=False, mask=False, weight=True, bias=False)
        tb = {}
        for name, needs in (('bwd_in', needs_i), ('bwd_w', needs_w)):
            for _ in range(2):
                dcn._backward(x, off, None, w, None, go, shape, packed, needs)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.iters):
                dcn._backward(x, off, None, w, None, go, shape, packed, needs)
            e1.record()
            torch.cuda.synchronize()
            tb[name] = e0.elapsed_time(e1) / args.iters * 1e-3
        e0.record()
        for _ in range(args.iters):
            dcn.pack_weight(w, shape)
        e1.record()
        torch.cuda.synchronize()
        tpack = e0.elapsed_time(e1) / args.iters * 1e-3
        rows.append(dict(B=B, H=H, W=W, k=k, us=round(t * 1e6, 1), bf16_us=round(tf['bf16'] * 1e6, 1),
                         exact_us=round(tf['exact'] * 1e6, 1), tflops=round(flops / t / 1e12, 1),
                         mfma_frac=round(flops / t / FP32_MFMA_PEAK, 3),
                         hbm_frac=round(byts / t / HBM_PEAK, 4),
                         bwd_in_us=round(tb['bwd_in'] * 1e6, 1), bwd_in_frac=round(flops / tb['bwd_in'] / FP32_MFMA_PEAK, 3),
                         bwd_w_us=round(tb['bwd_w'] * 1e6, 1), bwd_w_frac=round(flops / tb['bwd_w'] / FP32_MFMA_PEAK, 3),
                         pack_us=round(tpack * 1e6, 1)))
        print(json.dumps(rows[-1]), flush=True)


if __name__ == '__main__':
    main()
