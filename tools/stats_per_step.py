#!/usr/bin/env python
"""rocprofv3 --stats CSV of a bench.py run -> per-kernel ms per step (steps = launches of the loss kernel)."""
import csv
import glob
import sys

d = sys.argv[1]
f = sorted(glob.glob(d + "/stats/**/r_kernel_stats.csv", recursive=True))
rows = list(csv.DictReader(open(f[0])))
steps = next(int(r["Calls"]) for r in rows if r["Name"].startswith("void softmax_ce_tiled_kernel"))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"steps {steps}  kernel time per step {tot / steps / 1e6:.3f} ms")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    t, c = float(r["TotalDurationNs"]), int(r["Calls"])
    print(f"{t / steps / 1e6:7.3f} ms/step {c / steps:6.1f}/step avg {t / c / 1e3:8.1f} us  {r['Name'][:120]}")
