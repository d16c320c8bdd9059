#!/bin/bash
# Run ON THE GPU BOX: one rocprofv3 --pmc pass over one conv shape of tools/bench_kernels.py.
#   bash tools/pmc_run.sh <tag> <shape> <kinds> <name> <counter> [<counter> ...]
# Output: gpurun_out/pmc_<tag>/<name>_<shape>/ (csv) ; summarise with tools/pmc_table.py
set -e -o pipefail
TAG=$1; SHAPE=$2; KINDS=$3; NAME=$4; shift 4
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/${NAME}_$SHAPE" -o r -- \
    python3 "$R/tools/bench_kernels.py" --only "$SHAPE" --kinds "$KINDS" --iters 3 > "$OUT/${NAME}_$SHAPE.log" 2>&1
