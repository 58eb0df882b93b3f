#!/bin/bash
# GPU box: the randomised soaks on round 6's final library (SOAK_SECONDS each), results copied as gpurun_out/r6_soak_<name>.json
S=${SOAK_SECONDS:-240}
for name in msm fr provers psnark world dist_native; do
  SOAK_SECONDS=$S SOAK_SEED=${SOAK_SEED:-20260930} timeout $((S + 900)) python -m pytest tests/soak_$name.py -q -s 2>&1 | tail -4 > gpurun_out/r6_soak_$name.txt
  for f in gpurun_out/soak_$name*.json; do [ -f "$f" ] && cp "$f" gpurun_out/r6_$(basename $f); done
  echo "== $name"; tail -3 gpurun_out/r6_soak_$name.txt
done
