#!/usr/bin/env python
"""Same-box check of two libflairhip builds against each other and against torch on the BatchNorm entry points
(seeded inputs, several shapes, both dtypes):  python tools/bn_variant_check.py <tag>  writes gpurun_out/bn_check_<tag>.pt;
python tools/bn_variant_check.py --compare a b  prints the largest differences."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flair-for-aigle_amd"))
OUT = os.path.join(ROOT, "gpurun_out")

SHAPES = [(200, 32), (200, 16), (1000, 64), (4096, 128), (12345, 8), (2 * 37 * 41, 96), (65536, 64), (333, 256)]


def run(tag):
    from flairhip import ops
    dev = torch.device("cuda:0")
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        for npix, C in SHAPES:
            g = torch.Generator().manual_seed(npix * 131 + C)
            x = torch.randn(1, 1, npix, C, generator=g).to(dt).to(dev)
            dy = torch.randn(1, 1, npix, C, generator=g).to(dt).to(dev)
            gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), torch.randn(C, generator=g).to(dev)
            rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
            sc, sh, mean, rstd = ops.bn_stats(x, gamma, beta, rm, rv, 0.1, 1e-5)
            y = ops.bn_apply(x, sc, sh, None, True)
            y2 = ops.bn_apply(x, sc, sh, dy, False)
            dx, _, dg, db = ops.bn_bwd(x, dy, None, gamma, beta, mean, rstd, True, False)
            dx1, dres, dg1, db1 = ops.bn_bwd(x, dy, y, gamma, beta, mean, rstd, True, True)
            dx0, _, dg0, db0 = ops.bn_bwd(x, dy, None, gamma, beta, mean, rstd, False, False)
            s1, s2 = ops.channel_sums(x)
            # torch reference of the relu-from-x backward
            xf, dyf = x.float().reshape(-1, C), dy.float().reshape(-1, C)
            xf.requires_grad_(True)
            gp, bp = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
            yt = torch.relu(torch.nn.functional.batch_norm(xf, None, None, gp, bp, True, 0.1, 1e-5))
            yt.backward(dyf)
            key = f"{str(dt)[6:]}_{npix}x{C}"
            for n, t in (("sc", sc), ("sh", sh), ("y", y), ("y2", y2), ("dx", dx), ("dg", dg), ("db", db), ("dx1", dx1),
                         ("dres", dres), ("dg1", dg1), ("dx0", dx0), ("dg0", dg0), ("db0", db0), ("s1", s1), ("s2", s2)):
                res[f"{key}/{n}"] = t.float().cpu().flatten()
            res[f"{key}/T_dx"] = xf.grad.flatten().cpu()
            res[f"{key}/T_dg"] = gp.grad.cpu()
            res[f"{key}/T_db"] = bp.grad.cpu()
    torch.cuda.synchronize()
    os.makedirs(OUT, exist_ok=True)
    torch.save(res, os.path.join(OUT, f"bn_check_{tag}.pt"))


def compare(a, b):
    ra, rb = (torch.load(os.path.join(OUT, f"bn_check_{t}.pt")) for t in (a, b))
    worst = []
    for k in ra:
        if "/T_" in k:
            continue
        d = (ra[k] - rb[k]).abs().max().item() / max(1e-6, rb[k].abs().max().item())
        worst.append((d, k))
    worst.sort(reverse=True)
    print(f"{a} vs {b}: largest relative differences")
    for d, k in worst[:12]:
        print(f"  {d:.3e}  {k}")
    print(f"{a} vs torch (fp32 cases)")
    for k in ra:
        if k.startswith("float32") and k.endswith("/T_dx"):
            base = k[:-5]
            for n in ("dx", "dg", "db"):
                t, g = ra[f"{base}/T_{n}"], ra[f"{base}/{n}"]
                print(f"  {base:22s} {n}: {(t - g).abs().max().item() / max(1e-6, t.abs().max().item()):.2e}")


if __name__ == "__main__":
    if sys.argv[1] == "--compare":
        compare(sys.argv[2], sys.argv[3])
    else:
        run(sys.argv[1])
