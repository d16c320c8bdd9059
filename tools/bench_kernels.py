#!/usr/bin/env python
"""Micro-benchmark of single libflairhip kernels on the layer shapes of the batch-32 U-Net step.

  python tools/bench_kernels.py                 # table: shape, us, TFLOP/s or GB/s
  python tools/bench_kernels.py --only conv128  # one shape, many launches (use under rocprofv3 --pmc)

Inputs are random (never zeros: DVFS), each kernel is timed with HIP events on the stream it runs on,
median of several batches of back-to-back launches.
"""
from __future__ import annotations

import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flair-for-aigle_amd"))

import torch

from flairhip import ops

B = 32
# name, Cin, Cout, k, stride, pad, H (input), what
CONV_SHAPES = [
    ("conv64", 64, 64, 3, 1, 1, 128),
    ("conv128", 128, 128, 3, 1, 1, 64),
    ("conv256", 256, 256, 3, 1, 1, 32),
    ("conv512", 512, 512, 3, 1, 1, 16),
    ("dec0c1", 768, 256, 3, 1, 1, 32),
    ("dec1c1", 384, 128, 3, 1, 1, 64),
    ("dec2c1", 192, 64, 3, 1, 1, 128),
    ("dec3c1", 128, 32, 3, 1, 1, 256),
    ("dec3c2", 32, 32, 3, 1, 1, 256),
    ("dec4c1", 32, 16, 3, 1, 1, 512),
    ("dec4c2", 16, 16, 3, 1, 1, 512),
    ("head", 16, 19, 3, 1, 1, 512),
    ("l2down", 64, 128, 3, 2, 1, 128),
    ("stem", 5, 64, 7, 2, 3, 512),
]


def timeit(fn, iters=20, reps=7, warm_s=0.4):
    """median over `reps` batches of `iters` back-to-back launches, after `warm_s` seconds of the same launches (a
    fresh process starts at a low power state: the first batches read 15-20 % slow)"""
    import time
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    while time.time() - t0 < warm_s:
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
    best = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        best.append(s.elapsed_time(e) / iters * 1e3)
    best.sort()
    return best[len(best) // 2]


# name, C, H of the BatchNorm layers of the step (count per step in the comment)
BN_SHAPES = [
    ("bn_stem", 64, 256),    # 1
    ("bn_l1", 64, 128),      # 6 (+2 decoder)
    ("bn_l2", 128, 64),      # 9 (+2)
    ("bn_l3", 256, 32),      # 13 (+2)
    ("bn_l4", 512, 16),      # 7
    ("bn_d3", 32, 256),      # 2
    ("bn_d4", 16, 512),      # 2
]


def bench_bn(args):
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    print(f"{'shape':10s} {'kernel':10s} {'us':>9s} {'GB/s':>9s}")
    for name, C, H in BN_SHAPES:
        if args.only and args.only != name:
            continue
        x = torch.randn(B, H, H, C, device=dev).to(dt)
        dy = torch.randn(B, H, H, C, device=dev).to(dt)
        y = torch.empty_like(x)
        gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        sc, sh, mean, rstd = ops.bn_stats(x, gamma, beta, rm, rv, 0.1, 1e-5)
        nb = x.numel() * 2
        runs = [
            ("stats", lambda: ops.bn_stats(x, gamma, beta, rm, rv, 0.1, 1e-5), nb),
            ("apply", lambda: ops.bn_apply(x, sc, sh, None, True, out=y), 2 * nb),
            ("apply+res", lambda: ops.bn_apply(x, sc, sh, dy, True, out=y), 3 * nb),
            ("bwd(m2)", lambda: ops.bn_bwd(x, dy, None, gamma, beta, mean, rstd, True, False), 5 * nb),
            ("bwd(m1,dr)", lambda: ops.bn_bwd(x, dy, y, gamma, beta, mean, rstd, True, True), 8 * nb),
        ]
        for kname, fn, nbytes in runs:
            us = timeit(fn, iters=args.iters)
            print(f"{name:10s} {kname:10s} {us:9.1f} {nbytes / us / 1e3:9.0f}")
    if args.only:
        return
    # the other streaming kernels of the step (algorithmic bytes)
    x = torch.randn(B, 256, 256, 64, device=dev).to(dt)
    yp, idx = ops.maxpool3x3s2_fwd(x)
    dyp = torch.randn_like(yp)
    add = torch.randn_like(x)
    xin = torch.randn(B, 5, 512, 512, device=dev)
    logits = torch.randn(B, 512, 512, 32, device=dev).to(dt)
    tgt = torch.randint(0, 19, (B, 512, 512), device=dev, dtype=torch.uint8)
    cw = torch.tensor([1.0] * 15 + [0.0] * 4, device=dev)
    nb = x.numel() * 2
    runs = [
        ("maxpool", "fwd", lambda: ops.maxpool3x3s2_fwd(x), nb + nb // 4 + nb // 8),
        ("maxpool", "bwd+add", lambda: ops.maxpool3x3s2_bwd(dyp, idx, (256, 256), add), 2 * nb + nb // 4 + nb // 8),
        ("layout", "nchw->nhwc", lambda: ops.nchw_to_nhwc(xin, dt), xin.numel() * 4 + B * 512 * 512 * 16 * 2),
        ("loss", "softmax_ce", lambda: ops.softmax_ce(logits, tgt, cw, 19, want_grad=True, want_pred=True),
         2 * logits.numel() * 2 + 2 * tgt.numel()),
    ]
    for name, kname, fn, nbytes in runs:
        us = timeit(fn, iters=args.iters)
        print(f"{name:10s} {kname:10s} {us:9.1f} {nbytes / us / 1e3:9.0f}")


def bench_blas(args):
    """The equal-FLOP GEMM of every MFMA-bound conv shape (M = B*Ho*Wo pixels, N = Cout, K = Cin*k*k) through torch's
    matmul = hipBLASLt on this image, same box, random bf16 operands: what a tuned vendor GEMM reaches on these M/N/K,
    i.e. the ceiling the implicit-GEMM kernels are measured against (an im2col-free conv moves 1/9 of the A bytes, so
    it can be faster, but not by the MFMA schedule)."""
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    print(f"{'shape':10s} {'M':>8s} {'N':>5s} {'K':>6s} {'us':>9s} {'TFLOP/s':>9s}   layout")
    for name, cin, cout, k, stride, pad, H in CONV_SHAPES:
        if args.only and args.only != name:
            continue
        if cout < 64 or k != 3 or stride != 1:
            continue
        Ho = (H + 2 * pad - k) // stride + 1
        M, N, K = B * Ho * Ho, cout, cin * k * k
        a = torch.randn(M, K, device=dev).to(dt)
        w = torch.randn(N, K, device=dev).to(dt)
        wt = w.t().contiguous()
        for lay, fn in (("a[M,K] @ w[N,K]^T", lambda: torch.matmul(a, w.t())), ("a[M,K] @ w[K,N]", lambda: torch.matmul(a, wt))):
            us = timeit(fn, iters=args.iters)
            print(f"{name:10s} {M:8d} {N:5d} {K:6d} {us:9.1f} {2.0 * M * N * K / us / 1e6:9.1f}   {lay}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--blas", action="store_true", help="hipBLASLt (torch.matmul) on the equal-FLOP GEMM of each MFMA-bound conv shape")
    ap.add_argument("--kinds", default="fwd,dgrad,wgrad")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--bn", action="store_true", help="BatchNorm kernels instead of the convolutions")
    ap.add_argument("--data", default="randn", choices=["randn", "relu", "zeros"],
                    help="activation values: randn (worst case for power), relu (post-ReLU: half zeros, no "
                         "negative values -- what the network's 3x3 convs actually read), zeros")
    args = ap.parse_args()
    if args.bn:
        return bench_bn(args)
    if args.blas:
        return bench_blas(args)
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    kinds = args.kinds.split(",")
    print(f"{'shape':10s} {'kind':6s} {'us':>9s} {'TFLOP/s':>9s} {'GB/s(min traffic)':>18s}")
    for name, cin, cout, k, stride, pad, H in CONV_SHAPES:
        if args.only and args.only != name:
            continue
        cip = ops.pad_channels(cin)
        cop = 32 if cout == 19 else ops.pad_channels(cout)
        Ho = (H + 2 * pad - k) // stride + 1
        x = torch.randn(B, H, H, cip, device=dev)
        if args.data == "relu":
            x = x.relu()
        elif args.data == "zeros":
            x = x * 0
        x = x.to(dt)
        if cip > cin:
            x[..., cin:] = 0
        w = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
        dy = torch.randn(B, Ho, Ho, cop, device=dev).to(dt)
        flops = 2.0 * B * Ho * Ho * cout * cin * k * k
        bytes_min = (x.numel() + dy.numel()) * 2
        pw = ops.pack_conv_weight(w, dt, stride, cip)
        pwt = ops.pack_conv_weight(w, dt, stride, cop, transpose=True) if name != "stem" else None
        out = torch.empty(B, Ho, Ho, cop, device=dev, dtype=dt)
        dx = torch.empty(B, H, H, cip, device=dev, dtype=dt)
        dw = torch.empty(cout, cin, k, k, device=dev)
        stats = torch.empty(ops.conv_stat_rows(B, Ho, Ho, pw) * 2 * cop, device=dev)
        runs = {
            "fwd": lambda: ops.conv2d(x, pw, pad, cop, out=out),
            "fwdst": lambda: ops.conv2d(x, pw, pad, cop, out=out, stats=stats),  # + BatchNorm statistics epilogue
            "dgrad": lambda: ops.conv2d(dy, pwt, k - 1 - pad, cip, dil=stride, out_hw=(H, H), out=dx),
            "wgrad": lambda: ops.conv_wgrad(x, dy, cout, cin, k, k, stride, pad, out=dw),
        }
        for kind in kinds:
            if kind == "dgrad" and name == "stem":
                continue
            us = timeit(runs[kind], iters=args.iters)
            print(f"{name:10s} {kind:6s} {us:9.1f} {flops / us / 1e6:9.1f} {bytes_min / us / 1e3:18.0f}")


if __name__ == "__main__":
    main()
