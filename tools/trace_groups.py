#!/usr/bin/env python
"""Group a rocprofv3 --kernel-trace CSV by (kernel, grid size): calls, mean us, share -- separates the launches of one
symbol that serve different layer shapes.   python tools/trace_groups.py <..._kernel_trace.csv> [top]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
g = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0][-60:]
    g[(name, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in g.values())
print(f"total {tot / 1e6:.2f} ms over {len(rows)} launches")
for (name, blocks), v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print(f"{100 * sum(v) / tot:6.2f}%  {len(v):5d} x {sum(v) / len(v) / 1e3:8.1f} us  blocks {blocks:7d}  {name}")
