#!/bin/bash
# Run ON THE GPU BOX: bash tools/ab_stats.sh TAG "ENV=VAL ..."  -> rocprofv3 kernel stats of bench.py under that
# environment in gpurun_out/prof_TAG/ (per-kernel ms per step: tools/stats_per_step.py)
set -e -o pipefail
TAG=$1
shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o r -- \
    python3 "$R/bench.py" --no-cpu-baseline --no-extras > "$OUT/bench_stats.log" 2>&1
rm -f "$OUT"/stats/*/r_kernel_trace.csv "$OUT"/stats/r_kernel_trace.csv
python3 "$R/tools/stats_per_step.py" "$OUT" > "$OUT/per_step.txt"
