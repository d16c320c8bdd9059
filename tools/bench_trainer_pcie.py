#!/usr/bin/env python
"""PCIe-inclusive training throughput: HipTrainer.fit over HOST-resident batches (pinned), i.e. what a DataLoader hands
over, for the reference's batch schema (f32 imagery + f32 one-hot labels, 25 MB per tile) and for the compact schema
the product also accepts (uint8 imagery + '<MOD>_NORM' + uint8 class indices, 1.6 MB per tile).  bench.py's `value`
has its inputs resident in HBM; this is the number next to it.

  python tools/bench_trainer_pcie.py [--batch 32] [--steps 30]
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

MOD, TASK = "AERIAL_RGBI", "AERIAL_LABEL-COSIA"


class Cycle:
    """len()-able loader over a few pinned host batches, reused round-robin"""

    def __init__(self, batches, n):
        self.batches, self.n = batches, n

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            yield self.batches[i % len(self.batches)]


def run(schema: str, B: int, steps: int, warm: int = 10):
    from flairhip.configs import unet_resnet34_config
    from flair_hub.tasks.module_setup import build_segmentation_module
    from flair_hub.tasks.trainers import HipTrainer
    cfg = unet_resnet34_config(in_channels=5, precision="bf16", batch_size=B, total_steps=steps + warm + 8)
    torch.manual_seed(2025)
    task = build_segmentation_module(cfg, {MOD: 512}, "train")
    g = torch.Generator().manual_seed(1)
    batches = []
    for _ in range(3):
        t = torch.randint(0, 19, (B, 512, 512), generator=g)
        if schema == "reference":
            b = {MOD: torch.randn(B, 5, 512, 512, generator=g),
                 TASK: torch.nn.functional.one_hot(t, 19).permute(0, 3, 1, 2).float().contiguous()}
        else:
            b = {MOD: torch.randint(0, 255, (B, 5, 512, 512), generator=g, dtype=torch.uint8),
                 MOD + "_NORM": torch.tensor([[110.0] * 5, [50.0] * 5]), TASK: t.to(torch.uint8)}
        batches.append({k: v.pin_memory() for k, v in b.items()})
    mb = sum(v.numel() * v.element_size() for v in batches[0].values()) / 2 ** 20
    marks = {}
    orig = task.on_train_batch_end

    def hook(loss, batch, i):
        if task.global_step in (warm, warm + steps):
            torch.cuda.synchronize()
            marks[task.global_step] = time.perf_counter()
        return orig(loss, batch, i)

    task.on_train_batch_end = hook
    tr = HipTrainer(max_epochs=1, max_steps=warm + steps, hip_graph=True)
    tr.fit(task, train_dataloaders=Cycle(batches, warm + steps + 1))
    dt = marks[warm + steps] - marks[warm]
    return {"schema": schema, "host_MB_per_batch": round(mb, 1), "ms_per_step": round(dt / steps * 1e3, 3),
            "tiles_per_s": round(B * steps / dt, 1), "h2d_GBps_needed": round(mb / 1024 / (dt / steps), 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=30)
    args = ap.parse_args()
    for schema in ("compact", "reference"):
        print(json.dumps(run(schema, args.batch, args.steps)), flush=True)
    # raw H2D rate of this box for scale
    x = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
    d = torch.empty_like(x, device="cuda")
    d.copy_(x, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        d.copy_(x, non_blocking=True)
    torch.cuda.synchronize()
    print(json.dumps({"pinned_h2d_GBps": round(5 / (time.perf_counter() - t0), 1)}))


if __name__ == "__main__":
    main()
