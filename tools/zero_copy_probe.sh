#!/bin/bash
# GPU box: A/B of GM_ZERO_COPY (kernels write their small results straight into pinned host memory; bit 0 field paths, bit 1 MSM planes)
for rep in 1 2; do
for z in 0 1 2 3; do
  export GM_ZERO_COPY=$z
  echo "== GM_ZERO_COPY=$z (rep $rep)"
  python bench.py --steps 40 --warmup 10 --headline-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('msm20', d['value'], d['ms_per_step'])"
  python tools/run_snark.py -i 20 --repeat 9 --native | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('snark20', d['time_prover_s'], d['proof_sha256'][:12])"
  python tools/run_snark.py -i 24 --repeat 5 --native | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('snark24', d['time_prover_s'], d['proof_sha256'][:12])"
  python tools/run_psnark.py -i 20 --repeat 3 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('psnark20', d['time_prover_s'], d['proof_sha256'][:12])"
done
done
