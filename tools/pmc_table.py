#!/usr/bin/env python
"""Per-kernel means of every counter found under a gpurun_out/pmc_<tag>/ folder (tools/pmc_run.sh output).

  python tools/pmc_table.py gpurun_out/pmc_r02 [kernel-substring]
"""
import collections
import csv
import glob
import os
import sys


def main():
    folder = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else "conv"
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for path in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
        run = os.path.relpath(path, folder).split(os.sep)[0]
        for r in csv.DictReader(open(path)):
            if filt not in r["Kernel_Name"]:
                continue
            a = acc[(run, r["Kernel_Name"][:90])][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    for (run, k), cs in sorted(acc.items()):
        print(f"== {run}: {k}")
        for c, (s, n) in sorted(cs.items()):
            print(f"    {c:28s} {s / n:14.4e}  ({n} launches)")


if __name__ == "__main__":
    main()
