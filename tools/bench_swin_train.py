#!/usr/bin/env python
"""Training throughput of BASELINE.json's configs[3]: Swin-T encoder + UPerNet decoder (the reference's
`swin_tiny_patch4_window7_224-upernet`, models.monotemp_model.arch), 512 x 512 x 5 synthetic tiles, 19 COSIA classes,
bf16, AdamW + OneCycleLR, DropPath 0.1 -- SegmentationTask.training_step + backward + optimizer step on 1 MI355X.

  python tools/bench_swin_train.py [--arch swin_tiny_patch4_window7_224-upernet] [--batch 32] [--steps 20] [--graph]

Prints one JSON line (tiles/s, ms/step).  Not the driver's bench: bench.py stays on BASELINE.json's headline
configuration (configs[1], the U-Net).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flair-for-aigle_amd"))

import torch


def cpu_baseline(encoder: str, channels: int, tile: int, budget_s: float = 20.0, B: int = 2):
    """the oracle's training step (oracle/swin_upernet.py, fp32 eager torch, AdamW) on the host cores"""
    import torch.nn.functional as F
    sys.path.insert(0, ROOT)
    from oracle.swin_upernet import SwinUPerNet
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 16))
    torch.set_num_threads(threads)
    torch.manual_seed(2025)
    model = SwinUPerNet(encoder, channels, 19, tile).train()
    opt = torch.optim.AdamW(model.parameters(), lr=5e-5, weight_decay=0.01)
    x, t = torch.randn(B, channels, tile, tile), torch.randint(0, 19, (B, tile, tile))

    def step():
        loss = F.cross_entropy(model(x), t)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    t0 = time.perf_counter()
    step()
    warm = time.perf_counter() - t0
    n = max(1, min(4, int(budget_s / max(warm, 1e-3)) - 1))
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    dt = time.perf_counter() - t0
    return {"value": round(B * n / dt, 4), "unit": "tiles/s", "cores": threads, "kind": "port",
            "sample": f"{n} timed step(s) of batch {B} after 1 warm-up, oracle/swin_upernet.py (fp32 eager torch, AdamW)"}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="swin_tiny_patch4_window7_224-upernet")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--tile", type=int, default=512)
    ap.add_argument("--channels", type=int, default=5)
    ap.add_argument("--graph", action="store_true", help="capture the step as one hipGraph (GraphedTrainStep)")
    ap.add_argument("--cpu-baseline", action="store_true", help="also time the CPU oracle's step (about 20-40 s)")
    return ap.parse_args(argv)


def measure(args) -> dict:
    """one measurement; also called in-process by bench.py (extra object "swin_t" of the driver-run line)"""
    from flairhip.configs import unet_resnet34_config
    from flair_hub.tasks.module_setup import build_segmentation_module
    MOD, TASK = "AERIAL_RGBI", "AERIAL_LABEL-COSIA"
    dev = torch.device("cuda:0")
    cfg = unet_resnet34_config(in_channels=args.channels, precision="bf16", batch_size=args.batch,
                               total_steps=args.steps + args.warmup + 32)
    cfg["models"]["monotemp_model"]["arch"] = args.arch
    torch.manual_seed(2025)
    task = build_segmentation_module(cfg, {MOD: args.tile}, "train").to(dev)
    task.train()
    oc = task.configure_optimizers()
    optimizer, scheduler = oc["optimizer"], oc["lr_scheduler"]["scheduler"]
    g = torch.Generator(device=dev).manual_seed(7)
    B, S = args.batch, args.tile
    batch = {MOD: torch.randn(B, args.channels, S, S, generator=g, device=dev),
             TASK: torch.randint(0, 19, (B, S, S), generator=g, device=dev, dtype=torch.uint8)}
    nparams = sum(p.numel() for p in task.model.parameters())

    def eager_step(i):
        loss = task.training_step(batch, i)
        optimizer.zero_grad(set_to_none=True)
        loss.backward()
        optimizer.step()
        scheduler.step()
        return loss

    graphed = None
    if args.graph:
        from flairhip.graph import GraphedTrainStep
        graphed = GraphedTrainStep(task, optimizer, batch, warmup_steps=3, after_step=scheduler.step)
    else:
        for i in range(4):
            eager_step(i)
    step = (lambda i: graphed(batch)) if graphed is not None else eager_step
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # roofline pass (bench.py's convention): a few eager steps with HIP events around the MFMA GEMM launches
    from flairhip import ops
    recs = []
    orig_lin, orig_wg = ops.linear, ops.linear_wgrad

    def timed(sym, flops, fn, *a, **kw):
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        r = fn(*a, **kw)
        e_.record()
        recs.append((sym, flops, s_, e_))
        return r

    def lin(x, w, *a, **kw):
        M, K, N = x.numel() // x.shape[-1], x.shape[-1], w.shape[0]
        return timed("token GEMM fwd / dgrad (gemm_bf16_kernel, gemm256_bf16_kernel)", 2.0 * M * K * N, orig_lin, x, w, *a, **kw)

    def wg(x, dy, *a, **kw):
        M, K, N = x.numel() // x.shape[-1], x.shape[-1], dy.shape[-1]
        return timed("weight-gradient GEMM (gemm_tn_bf16_kernel + reduce)", 2.0 * M * K * N, orig_wg, x, dy, *a, **kw)

    ops.linear, ops.linear_wgrad = lin, wg
    import flairhip.swin as _sw
    _sw.ops = ops
    for i in range(2):
        eager_step(args.warmup + args.steps + i)
    torch.cuda.synchronize()
    ops.linear, ops.linear_wgrad = orig_lin, orig_wg
    agg = {}
    for sym, fl, s_, e_ in recs:
        a = agg.setdefault(sym, [0.0, 0.0, 0])
        a[0] += fl
        a[1] += s_.elapsed_time(e_) * 1e-3
        a[2] += 1
    dom = max(agg.items(), key=lambda kv: kv[1][1]) if agg else None
    roofline = None
    if dom:
        ach = dom[1][0] / dom[1][1] / 1e12
        roofline = {"bound": "mfma", "kernel": dom[0], "achieved": round(ach, 1), "peak": 2500.0, "unit": "TFLOP/s",
                    "frac": round(ach / 2500.0, 4), "launches_per_step": dom[1][2] / 2,
                    "avg_launch_ms": round(dom[1][1] / dom[1][2] * 1e3, 4),
                    "share_of_step": round(dom[1][1] / 2 / (dt / args.steps), 4)}
    cpu = None
    if args.cpu_baseline:
        cpu = cpu_baseline(args.arch.split("-")[0], args.channels, S)
    return ({
        "roofline": roofline, "cpu_baseline": cpu,
        "metric": f"{S}x{S}x{args.channels} tiles/sec (train fwd+bwd+AdamW), {args.arch}, 19 classes",
        "value": round(B * args.steps / dt, 1), "unit": "tiles/s", "ms_per_step": round(dt / args.steps * 1e3, 2),
        "batch": B, "dtype": "bf16", "mode": "hipGraph" if graphed is not None else "eager",
        "parameters": nparams, "loss": round(float(loss.detach()), 4), "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1),
        "data": "synthetic"})


def main():
    print(json.dumps(measure(parse())))


if __name__ == "__main__":
    main()
