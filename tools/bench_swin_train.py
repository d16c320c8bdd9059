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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="swin_tiny_patch4_window7_224-upernet")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--tile", type=int, default=512)
    ap.add_argument("--channels", type=int, default=5)
    ap.add_argument("--graph", action="store_true", help="capture the step as one hipGraph (GraphedTrainStep)")
    args = ap.parse_args()
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
    print(json.dumps({
        "metric": f"{S}x{S}x{args.channels} tiles/sec (train fwd+bwd+AdamW), {args.arch}, 19 classes",
        "value": round(B * args.steps / dt, 1), "unit": "tiles/s", "ms_per_step": round(dt / args.steps * 1e3, 2),
        "batch": B, "dtype": "bf16", "mode": "hipGraph" if graphed is not None else "eager",
        "parameters": nparams, "loss": round(float(loss), 4), "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1),
        "data": "synthetic"}))


if __name__ == "__main__":
    main()
