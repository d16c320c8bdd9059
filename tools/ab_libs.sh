#!/bin/bash
# Run ON THE GPU BOX: same-box A/B of libflairhip variants (tools/build_variant.py) on the whole training step.
#   bash tools/ab_libs.sh <reps> main <variant> [<variant> ...]      -> gpurun_out/ab_<variant>_<rep>.log, summary on stdout
set -e
REPS=$1; shift
V=flair-for-aigle_amd/csrc/build
for rep in $(seq 1 $REPS); do
for n in "$@"; do
  if [ $n = main ]; then unset FLAIRHIP_LIB; else export FLAIRHIP_LIB=$V/libflairhip_$n.so; fi
  python bench.py --no-cpu-baseline --steps 40 --warmup 5 > gpurun_out/ab_${n}_$rep.log 2>gpurun_out/ab_${n}_$rep.err
  python - <<PY
import json
l=[x for x in open('gpurun_out/ab_${n}_$rep.log') if x.startswith('{')][-1]
d=json.loads(l); print('$n', $rep, d['ms_per_step'])
PY
done
done
