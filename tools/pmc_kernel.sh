#!/bin/bash
# Run ON THE GPU BOX: SQ counters of one shape / kind of tools/bench_kernels.py, two passes (the SQ block holds 8 counters).
#   bash tools/pmc_kernel.sh r03 conv128 wgrad     ->  gpurun_out/pmc_r03/sq_conv128_wgrad_{a,b}/
set -e -o pipefail
TAG=${1:-r03}; SHAPE=${2:-conv128}; KIND=${3:-wgrad}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/sq_${SHAPE}_${KIND}_a" -o r -- python3 "$R/tools/bench_kernels.py" --only "$SHAPE" --kinds "$KIND" --iters 5 > "$OUT/sq_${SHAPE}_${KIND}_a.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS \
    --output-format csv -d "$OUT/sq_${SHAPE}_${KIND}_b" -o r -- python3 "$R/tools/bench_kernels.py" --only "$SHAPE" --kinds "$KIND" --iters 5 > "$OUT/sq_${SHAPE}_${KIND}_b.log" 2>&1
python3 - "$OUT/sq_${SHAPE}_${KIND}_a" "$OUT/sq_${SHAPE}_${KIND}_b" <<'PY' > "$OUT/sq_${SHAPE}_${KIND}.txt"
import csv, glob, sys, collections
for d in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        if "at::" in k or "elementwise" in k:
            continue
        print(k)
        for n, v in sorted(c.items()):
            print(f"    {n:28s} {sum(v) / len(v):16.0f}  (x{len(v)})")
PY
cat "$OUT/sq_${SHAPE}_${KIND}.txt"
