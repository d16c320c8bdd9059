#!/usr/bin/env python
"""Training throughput of the multi-modality / multi-task path: aerial 5-channel tiles
at 512 x 512 + DEM elevation 2-channel tiles at 128 x 128 (the 1 m DEM of a 0.2 m tile is bilinearly aligned per stage
by FusionHandler), two ResNet-34 encoders fused per stage by 1x1 convolutions, COSIA (19 classes) + LPIS (23 classes)
decoders with task weights 1 / 0.5 and an auxiliary aerial decoder -- everything on the same HIP kernels as the
mono-modal step of bench.py.

  python tools/bench_fusion.py [--batch 16] [--steps 20] [--warmup 5] [--no-graph]

Prints one JSON line (tiles/s, ms/step, graph or eager).  Not the driver's bench: bench.py stays on BASELINE.json's
headline configuration.
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


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--tile", type=int, default=512)
    ap.add_argument("--dem-tile", type=int, default=128)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--sentinel", type=int, default=0, metavar="T",
                    help="add the Sentinel-2 time-series branch (U-TAE, T dates of 10 x 10 px x 10 bands): BASELINE configs[4]")
    return ap.parse_args(argv)


def measure(args) -> dict:
    """one measurement; also called in-process by bench.py (extra object "fusion_sentinel" of the driver-run line)"""
    from flairhip.configs import fusion_unet_config
    from flair_hub.tasks.module_setup import build_segmentation_module
    MOD, DEM, COSIA, LPIS = "AERIAL_RGBI", "DEM_ELEV", "AERIAL_LABEL-COSIA", "ALL_LABEL-LPIS"
    dev = torch.device("cuda:0")
    cfg = fusion_unet_config(precision="bf16", batch_size=args.batch)
    cfg["hyperparams"]["total_steps"] = args.steps + args.warmup + 32
    sizes = {MOD: args.tile, DEM: args.dem_tile}
    if args.sentinel:
        cfg["modalities"]["inputs"]["SENTINEL2_TS"] = True
        cfg["modalities"]["inputs_channels"]["SENTINEL2_TS"] = list(range(1, 11))
        sizes["SENTINEL2_TS"] = 10
    torch.manual_seed(2025)
    task = build_segmentation_module(cfg, sizes, "train").to(dev)
    task.train()
    oc = task.configure_optimizers()
    optimizer, scheduler = oc["optimizer"], oc["lr_scheduler"]["scheduler"]
    g = torch.Generator(device=dev).manual_seed(7)
    B, S, D = args.batch, args.tile, args.dem_tile
    batch = {MOD: torch.randn(B, 5, S, S, generator=g, device=dev),
             DEM: torch.randn(B, 2, D, D, generator=g, device=dev),
             COSIA: torch.randint(0, 19, (B, S, S), generator=g, device=dev, dtype=torch.uint8),
             LPIS: torch.randint(0, 23, (B, S, S), generator=g, device=dev, dtype=torch.uint8)}
    if args.sentinel:
        T = args.sentinel
        batch["SENTINEL2_TS"] = torch.randn(B, T, 10, 10, 10, generator=g, device=dev)
        batch["SENTINEL2_DATES"] = torch.sort(torch.randint(0, 365, (B, T), generator=g, device=dev), dim=1).values.float()
    nparams = sum(p.numel() for p in task.model.parameters())

    def eager_step(i):
        loss = task.training_step(batch, i)
        optimizer.zero_grad(set_to_none=True)
        loss.backward()
        optimizer.step()
        scheduler.step()
        return loss

    graphed = None
    if not args.no_graph:
        from flairhip.graph import GraphedTrainStep
        graphed = GraphedTrainStep(task, optimizer, batch, warmup_steps=3, after_step=scheduler.step)
    else:
        for i in range(8):  # allocator steady state, see bench.py
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
    return ({
        "metric": "512x512x5 (+128x128x2 DEM" + (f" + {args.sentinel} Sentinel-2 dates" if args.sentinel else "") +
                  ") tiles/sec, two-encoder fused U-Net ResNet-34" + (" + U-TAE" if args.sentinel else "") +
                  ", COSIA + LPIS + aux decoder, train fwd+bwd+AdamW",
        "value": round(B * args.steps / dt, 2), "unit": "tiles/s", "ms_per_step": round(dt / args.steps * 1e3, 3),
        "batch": B, "steps": args.steps, "warmup": args.warmup, "dtype": "bf16", "hip_graph": graphed is not None,
        "parameters": nparams, "final_loss": round(float(loss.item()), 5),
        "peak_memory_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)})


def main():
    print(json.dumps(measure(parse())))


if __name__ == "__main__":
    main()
