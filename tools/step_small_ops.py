#!/usr/bin/env python
"""Which torch ops of one eager training step (bench.py's configuration) issue the small device copies / fills / adds
that sit between the library's kernels: torch.profiler table of aten::copy_ / clone / zeros / fill_ / add / mul with
input shapes and the Python call sites that reach them.

  python tools/step_small_ops.py [--batch 8]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "flair-for-aigle_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
args = ap.parse_args()

from flair_hub.tasks.module_setup import build_segmentation_module  # noqa: E402
from flairhip.configs import unet_resnet34_config  # noqa: E402

MOD, TASK = "AERIAL_RGBI", "AERIAL_LABEL-COSIA"
dev = torch.device("cuda:0")
cfg = unet_resnet34_config(in_channels=5, precision="bf16", batch_size=args.batch, total_steps=64)
torch.manual_seed(cfg["hyperparams"]["seed"])
task = build_segmentation_module(cfg, {MOD: 512}, "train").to(dev)
task.train()
oc = task.configure_optimizers()
opt, sched = oc["optimizer"], oc["lr_scheduler"]["scheduler"]
g = torch.Generator(device=dev).manual_seed(1)
batch = {MOD: torch.randn(args.batch, 5, 512, 512, generator=g, device=dev),
         TASK: torch.randint(0, 19, (args.batch, 512, 512), generator=g, device=dev, dtype=torch.uint8)}


def step(i):
    loss = task.training_step(batch, i)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    sched.step()


for i in range(3):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(3)
    torch.cuda.synchronize()
want = ("aten::copy_", "aten::clone", "aten::zeros", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::mul",
        "aten::mul_", "aten::_to_copy", "aten::sum", "aten::div", "aten::div_", "aten::ones", "aten::empty_like")
rows = {}
for ev in prof.events():
    if ev.name in want and ev.device_time_total > 0:
        site = next((s for s in ev.stack if "/repo/" in s and "tools/step_small_ops" not in s), ev.stack[0] if ev.stack else "?")
        key = (ev.name, str(ev.input_shapes)[:60], site.split("/repo/")[-1][:90])
        r = rows.setdefault(key, [0, 0.0])
        r[0] += 1
        r[1] += ev.device_time_total
for (name, shapes, site), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{n:3d} x {us / max(n, 1):7.1f} us  {name:14s} {shapes:60s} {site}")
