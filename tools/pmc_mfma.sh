#!/bin/bash
# Run ON THE GPU BOX: SQ counters of one conv shape (MFMA utilisation of the dominant kernel).  bash tools/pmc_mfma.sh r01 conv128
set -e -o pipefail
TAG=${1:-r01}; SHAPE=${2:-conv128}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/sq_$SHAPE" -o r -- python3 "$R/tools/bench_kernels.py" --only "$SHAPE" --kinds fwd --iters 5 > "$OUT/sq_$SHAPE.log" 2>&1
ls "$OUT/sq_$SHAPE"
