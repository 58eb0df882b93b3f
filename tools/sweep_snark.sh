#!/bin/bash
# time_prover seconds over INSTANCE_LOGSIZE (BASELINE.json north_star: 20-28), one process per size so
# a failure at a large size does not lose the smaller ones.  Output: one JSON line per size.
out=${1:-gpurun_out/time_prover_sweep.jsonl}
: > "$out"
for i in ${SIZES:-20 22 24 26 27 28}; do
  timeout ${PER_SIZE_TIMEOUT:-600} python tools/run_snark.py -i $i --repeat 2 >> "$out" 2>gpurun_out/sweep_err_$i.log || echo "{\"logn\": $i, \"error\": \"rc=$?\"}" >> "$out"
done
cat "$out"
