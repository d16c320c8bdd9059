set -e
V=flair-for-aigle_amd/csrc/build
for rep in 1 2; do
for n in main u4s u2c u1 old; do
  if [ $n = main ]; then unset FLAIRHIP_LIB; else export FLAIRHIP_LIB=$V/libflairhip_$n.so; fi
  python bench.py --no-cpu-baseline --steps 40 --warmup 5 > gpurun_out/ab_${n}_$rep.log 2>gpurun_out/ab_${n}_$rep.err
  python - <<PY
import json
l=[x for x in open('gpurun_out/ab_${n}_$rep.log') if x.startswith('{')][-1]
d=json.loads(l); print('$n', $rep, d['ms_per_step'])
PY
done
done
