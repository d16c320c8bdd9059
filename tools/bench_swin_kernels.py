#!/usr/bin/env python
"""Per-shape timings of the Swin-B / UPerNet kernels at the fork's zonal configuration (512 px tiles, window 12).

  python tools/bench_swin_kernels.py [--batch 8] [--kinds linear,attn,ln]

linear: every token GEMM shape of Swin-B (M = B * map^2), TFLOP/s and the HBM time of its operands at 5 TB/s;
attn:   ffa_window_attention per stage; ln: ffa_layer_norm per stage.  Warm-up first (clocks), HIP-event timing.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flair-for-aigle_amd"))

import torch

from flairhip import ops


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t_end = time.time() + 0.3
    while time.time() < t_end:
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--kinds", default="linear,attn,ln")
    ap.add_argument("--blas", action="store_true", help="also time torch's (hipBLASLt) matmul on the same shapes")
    ap.add_argument("--variant", default="base", choices=["base", "tiny"], help="Swin-B/w12 or Swin-T/w7 shapes")
    ap.add_argument("--bwd", action="store_true", help="also time the weight-gradient GEMM and the attention backward")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B = args.batch
    stages = [(128, 128, 4), (64, 256, 8), (32, 512, 16), (16, 1024, 32)]  # map size, C, heads
    win = 12
    if args.variant == "tiny":
        stages, win = [(128, 96, 3), (64, 192, 6), (32, 384, 12), (16, 768, 24)], 7
    kinds = args.kinds.split(",")
    if "linear" in kinds:
        print(f"{'shape':34s} {'us':>8s} {'TFLOP/s':>8s} {'HBM us@5TB/s':>12s}" + ("  hipBLASLt us  TF" if args.blas else ""))
        for si, (S, C, _) in enumerate(stages):
            M = B * S * S
            shapes = [("qkv", C, 3 * C, 0, False), ("proj", C, C, 0, True), ("fc1", C, 4 * C, 1, False),
                      ("fc2", 4 * C, C, 0, True)]
            if si > 0:
                shapes.append(("merge", 2 * C, C, 0, False))
            for name, K, N, act, res in shapes:
                x = torch.randn(M, K, device=dev).to(torch.bfloat16)
                w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
                b = torch.randn(N, device=dev)
                r = torch.randn(M, N, device=dev).to(torch.bfloat16) if res else None
                out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
                dt = timeit(lambda: ops.linear(x, w, b, act=act, residual=r, out=out))
                fl = 2.0 * M * K * N
                byt = 2.0 * (M * K + M * N * (2 if res else 1) + N * K)
                line = f"s{si + 1} {name:6s} M={M:7d} K={K:5d} N={N:5d} {dt * 1e6:8.1f} {fl / dt / 1e12:8.0f} {byt / 5e12 * 1e6:12.1f}"
                if args.blas:
                    dt2 = timeit(lambda: torch.matmul(x, w.t()))
                    line += f"   {dt2 * 1e6:8.1f} {fl / dt2 / 1e12:5.0f}"
                if args.bwd:
                    dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
                    dt3 = timeit(lambda: ops.linear_wgrad(x, dy, with_bias=True))
                    line += f"   wgrad {dt3 * 1e6:8.1f} us {fl / dt3 / 1e12:5.0f} TF"
                print(line)
    if "attn" in kinds:
        for si, (S, C, heads) in enumerate(stages):
            ws = min(win, S)
            qkv = torch.randn(B, S, S, 3 * C, device=dev).to(torch.bfloat16)
            bias = torch.randn(3 * C, device=dev)
            table = torch.randn((2 * ws - 1) ** 2, heads, device=dev)
            for shift in (0, ws // 2 if S > ws else 0):
                dt = timeit(lambda: ops.window_attention(qkv, bias, table, heads, ws, shift, 32 ** -0.5))
                byt = 2.0 * B * S * S * 4 * C
                extra = ""
                if args.bwd:
                    dout = torch.randn(B, S, S, C, device=dev).to(torch.bfloat16)
                    dtb = timeit(lambda: ops.window_attention_bwd(qkv, dout, bias, table, heads, ws, shift, 32 ** -0.5))
                    extra = f"   bwd {dtb * 1e6:8.1f} us"
                print(f"s{si + 1} attention map {S:3d} C={C:4d} heads {heads:2d} ws {ws} shift {shift}: {dt * 1e6:8.1f} us"
                      f"  (HBM {byt / 5e12 * 1e6:6.1f} us @5TB/s)" + extra)
    if "ln" in kinds:
        for si, (S, C, _) in enumerate(stages):
            x = torch.randn(B, S, S, C, device=dev).to(torch.bfloat16)
            g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
            dt = timeit(lambda: ops.layer_norm(x, g, b))
            byt = 4.0 * B * S * S * C
            print(f"s{si + 1} layer_norm rows {B * S * S:7d} C={C:4d}: {dt * 1e6:8.1f} us  ({byt / dt / 1e12:5.2f} TB/s)")


if __name__ == "__main__":
    main()
