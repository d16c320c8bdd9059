#!/usr/bin/env python
"""Where does a data-parallel hipGraph step spend its time?  Phase timings (host wall clock with a device synchronise
after every phase) of GraphedTrainStep(grad_reduce=...) on the bench model:

  FFA_BENCH_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 \
      tools/ddp_graph_probe.py --batch 8          # two ranks sharing one GPU (rehearsal)
  python tools/ddp_graph_probe.py --batch 8       # one rank, one-rank gloo group (always_sync)

Written to explain the 13 s/step of round 2's `--ddp-graph` rehearsal (gpurun_out/n2g.err).
"""
from __future__ import annotations

import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "flair-for-aigle_amd")):
    sys.path.insert(0, _p)

import torch
import torch.distributed as dist

TASK = "AERIAL_LABEL-COSIA"
MOD = "AERIAL_RGBI"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=6)
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    backend = os.environ.get("FFA_BENCH_BACKEND", "gloo")
    dev = torch.device("cuda", 0 if backend == "gloo" else int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29641")
        dist.init_process_group(backend, rank=0, world_size=1)
    else:
        dist.init_process_group(backend)

    from flairhip.configs import unet_resnet34_config
    from flairhip.distributed import GradSync
    from flairhip.graph import GraphedTrainStep
    from flair_hub.tasks.module_setup import build_segmentation_module

    cfg = unet_resnet34_config(in_channels=5, precision="bf16", batch_size=args.batch, total_steps=200)
    torch.manual_seed(cfg["hyperparams"]["seed"])
    task = build_segmentation_module(cfg, {MOD: 512}, "train").to(dev)
    task.train()
    oc = task.configure_optimizers()
    optimizer, scheduler = oc["optimizer"], oc["lr_scheduler"]["scheduler"]
    g = torch.Generator(device=dev).manual_seed(2025 + rank)
    B = args.batch
    batch = {MOD: torch.randn(B, 5, 512, 512, generator=g, device=dev),
             TASK: torch.randint(0, 19, (B, 512, 512), generator=g, device=dev, dtype=torch.uint8)}

    def stamp(label, t0):
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rank == 0:
            print(f"[probe] {label:38s} {dt * 1e3:10.2f} ms", file=sys.stderr, flush=True)
        return time.perf_counter()

    t0 = time.perf_counter()
    sync = GradSync(task.model, hooks=False, broadcast_from_rank0=False, always_sync=(world == 1))
    phases = {}

    def reduce_probe(params, grads):
        t = time.perf_counter()
        sync.reduce_grads(params, grads)
        torch.cuda.synchronize()
        phases["reduce"] = time.perf_counter() - t

    graphed = GraphedTrainStep(task, optimizer, batch, warmup_steps=2, after_step=scheduler.step, grad_reduce=reduce_probe)
    t0 = stamp("construct (2 eager steps + capture)", t0)
    for i in range(args.steps):
        t1 = time.perf_counter()
        graphed.graph.replay()
        torch.cuda.synchronize()
        t_replay = time.perf_counter() - t1
        t2 = time.perf_counter()
        sync.reduce_grads(graphed.params, graphed.static_grads)
        torch.cuda.synchronize()
        t_red = time.perf_counter() - t2
        t3 = time.perf_counter()
        optimizer.step()
        torch.cuda.synchronize()
        t_opt = time.perf_counter() - t3
        if rank == 0:
            print(f"[probe] step {i}: replay {t_replay * 1e3:9.2f}  reduce {t_red * 1e3:9.2f}  optimizer {t_opt * 1e3:9.2f} ms",
                  file=sys.stderr, flush=True)
    # the same through the public call
    for i in range(3):
        t1 = time.perf_counter()
        graphed(batch)
        torch.cuda.synchronize()
        if rank == 0:
            print(f"[probe] graphed() call {i}: {(time.perf_counter() - t1) * 1e3:9.2f} ms (reduce inside: "
                  f"{phases.get('reduce', 0) * 1e3:.2f})", file=sys.stderr, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
