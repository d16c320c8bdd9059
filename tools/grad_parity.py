#!/usr/bin/env python
"""Per-parameter gradient error of the product (fp32 or bf16 mode) against the oracle in float32 AND float64, in
backward order -- to tell conditioning (oracle f32 vs f64 differs as much) from a kernel that is off.

  python tools/grad_parity.py [--tile 512] [--precision fp32]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "flair-for-aigle_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import copy

import torch
import torch.nn.functional as F

from helpers import MOD, TASK, make_pair


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tile", type=int, default=512)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--seed", type=int, default=17)
    args = ap.parse_args()
    task, oracle, _ = make_pair(precision=args.precision)
    g = torch.Generator().manual_seed(args.seed)
    x = torch.randn(2, 5, args.tile, args.tile, generator=g)
    t = torch.randint(0, 19, (2, args.tile, args.tile), generator=g)
    w = torch.tensor([1.0] * 15 + [0.0] * 4)
    o64 = copy.deepcopy(oracle).double().train()
    oracle.train()
    l32 = F.cross_entropy(oracle(x), t, weight=w)
    l32.backward()
    l64 = F.cross_entropy(o64(x.double()), t, weight=w.double())
    l64.backward()
    task.train()
    loss, _, _ = task.step({MOD: x.cuda(), TASK: t.cuda()}, training=True)
    loss.backward()
    torch.cuda.synchronize()
    print(f"loss product {loss.item():.7f}  oracle f32 {l32.item():.7f}  oracle f64 {l64.item():.9f}")
    g32, g64 = dict(oracle.named_parameters()), dict(o64.named_parameters())
    rows = []
    for name, p in task.model.named_parameters():
        if name.startswith("fusion_handler."):
            continue
        ok = ("encoder." if name.startswith("encoders.") else "") + name.split(".seg_model.", 1)[1]
        r64 = g64[ok].grad
        e_prod = ((p.grad.double().cpu() - r64).norm() / r64.norm()).item()
        e_o32 = ((g32[ok].grad.double() - r64).norm() / r64.norm()).item()
        rows.append((ok, e_prod, e_o32))
    print(f"{'parameter':48s} {'product vs f64':>15s} {'oracle f32 vs f64':>18s}")
    for ok, a, b in rows:
        print(f"{ok:48s} {a:15.3e} {b:18.3e}")


if __name__ == "__main__":
    main()
