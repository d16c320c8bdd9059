#!/usr/bin/env python
"""Turn the raw rocprofv3 output of tools/collect_profiles.sh (gpurun_out/prof_<tag>/) into the committed summaries:

  profiles/<tag>_bench_kernel_stats.csv        rocprofv3 --kernel-trace --stats of `python3 bench.py --no-cpu-baseline`
  profiles/<tag>_pmc_fetch_write_per_launch.txt  per-kernel FETCH_SIZE / WRITE_SIZE means
  profiles/<tag>_pmc_traffic.json               HBM bytes per launch, read by bench.py's roofline.traffic

HBM bytes per launch = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024: on gfx950 FETCH_SIZE reports half of the bytes
of wide (16 B per lane) streaming reads, WRITE_SIZE is exact for 16-B stores (MI355X_MICROARCH.md, HBM section).

  python tools/pmc_summary.py r01
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counter_means(folder, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for path in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            a = acc[r["Kernel_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
    fetch, write = counter_means(os.path.join(src, "fetch"), "FETCH_SIZE"), counter_means(os.path.join(src, "write"), "WRITE_SIZE")
    kernels = {}
    lines = [f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python3 bench.py --steps 2 --warmup 1 "
             f"--no-cpu-baseline --no-graph`",
             "FETCH_SIZE, WRITE_SIZE in KB per launch (mean); FETCH x2 correction for gfx950 wide streaming reads "
             "(MI355X_MICROARCH.md, HBM section)",
             "kernel | launches | FETCH KB raw | FETCH MB corrected | WRITE KB | WRITE MB"]
    order = sorted(fetch, key=lambda k: -(2 * fetch[k][0] + write.get(k, (0, 0))[0]) * fetch[k][1])
    for k in order:
        f, n = fetch[k]
        w = write.get(k, (0.0, 0))[0]
        kernels[k] = {"launches": n, "fetch_kb_raw": f, "write_kb": w, "hbm_bytes_per_launch": 2 * f * 1024 + w * 1024}
        lines.append(f"{k[:70]:70s} {n:5d} {f:12.0f} {2 * f * 1024 / 1e6:9.1f} {w:12.0f} {w * 1024 / 1e6:9.1f}")
    note = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes over `python3 bench.py --steps 2 --warmup 1 "
            "--no-cpu-baseline --no-graph`; hbm_bytes_per_launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE "
            "reads half of a wide streaming read: MI355X_MICROARCH.md, HBM)")
    json.dump({"note": note, "kernels": kernels}, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    open(os.path.join(dst, f"{tag}_pmc_fetch_write_per_launch.txt"), "w").write("\n".join(lines[:40]) + "\n")
    print(f"wrote profiles/{tag}_*: {len(kernels)} kernels with counters, stats file {'copied' if stats else 'MISSING'}")


if __name__ == "__main__":
    main()
