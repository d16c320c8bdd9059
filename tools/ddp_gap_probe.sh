#!/bin/bash
# Run ON THE GPU BOX: kernel timeline of the data-parallel graph step in a one-rank RCCL group -- where the +0.4 ms over the
# whole-step graph sit (gaps around the collective and the optimizer graph).  bash tools/ddp_gap_probe.sh
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/ddp_gap
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export FFA_BENCH_FORCE_DDP=1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o r -- python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-extras > "$OUT/bench.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# find adamw kernels: gap before the first adamw of a step and after the last one; and the nccl kernel
idx = [i for i, r in enumerate(rows) if "adamw_multi" in r["Kernel_Name"]]
import statistics
gaps_before, gaps_after, ar = [], [], []
for i in idx:
    if i and "adamw_multi" not in rows[i - 1]["Kernel_Name"]:
        # walk back over the non-graph kernels between graph A's last kernel and this one
        j = i - 1
        chain = []
        while j > 0 and len(chain) < 8:
            chain.append((rows[j]["Kernel_Name"][:40], int(rows[j]["End_Timestamp"]) - int(rows[j]["Start_Timestamp"]),
                          int(rows[j + 1]["Start_Timestamp"]) - int(rows[j]["End_Timestamp"])))
            j -= 1
        gaps_before.append(chain)
for c in gaps_before[5:8]:
    print("--- kernels before the optimizer graph (name, duration ns, gap to next ns), newest first")
    for x in c:
        print("   ", x)
PY
