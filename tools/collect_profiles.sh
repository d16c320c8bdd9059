#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repository root:  bash tools/collect_profiles.sh r01
# Three separate rocprofv3 runs of bench.py (timing and counters are never mixed; FETCH_SIZE and WRITE_SIZE do not
# fit one pass -- MI355X_MICROARCH.md, counters table).  Raw output goes to gpurun_out/ (scratch); the summaries the
# judge reads are written by tools/pmc_summary.py into profiles/ on the build side.
set -e -o pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o r -- \
    python3 "$R/bench.py" --no-cpu-baseline --no-extras > "$OUT/bench_stats.log" 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o r -- \
    python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-graph > "$OUT/bench_fetch.log" 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o r -- \
    python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-graph > "$OUT/bench_write.log" 2>&1
rm -f "$OUT"/*/r_kernel_trace.csv  # large; the stats file carries what is committed
ls -la "$OUT"/*
