#!/bin/bash
# GPU box: A/B of one environment switch over the headline MSM and three provers.  Usage: tools/ab_env.sh VAR "v0 v1 ..." [reps]
VAR=$1; VALS=$2; REPS=${3:-2}
for rep in $(seq $REPS); do
for z in $VALS; do
  export $VAR=$z
  m=$(python bench.py --steps 40 --warmup 10 --headline-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'])")
  a=$(python tools/run_snark.py -i 20 --repeat 9 --native 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['time_prover_s'], d['proof_sha256'][:8])")
  b=$(python tools/run_snark.py -i 24 --repeat 5 --native 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['time_prover_s'], d['proof_sha256'][:8])")
  c=$(python tools/run_psnark.py -i 20 --repeat 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['time_prover_s'], d['proof_sha256'][:8])")
  echo "$VAR=$z rep $rep | msm20 $m | snark20 $a | snark24 $b | psnark20 $c"
done
done
