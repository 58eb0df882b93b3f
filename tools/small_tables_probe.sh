#!/bin/bash
# GPU box: GM_SMALL_TABLE_C experiment -- one-call MSMs of 2^9 .. 2^17 pairs against a resident key, and the small provers
run() {
  echo "== GM_SMALL_TABLE_C=$GM_SMALL_TABLE_C GM_FLAT_TABLE_MAX=$GM_FLAT_TABLE_MAX GM_SMALL_TABLE_MIN=$GM_SMALL_TABLE_MIN"
  python tools/msm_sizes.py 2>/dev/null | python -c "
import sys, json
r = {json.loads(l)['logn']: json.loads(l)['ms'] for l in sys.stdin if l.startswith('{')}
print('one-call ms:', ' '.join(f'2^{k}:{r[k]}' for k in range(9, 19)))"
  for i in 18 20; do
    python tools/run_snark.py -i $i --repeat 9 --native 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('snark$i', d['time_prover_s'], d['proof_sha256'][:8])"
  done
  python tools/run_psnark.py -i 18 --repeat 5 --native 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('psnark18', d['time_prover_s'], d['proof_sha256'][:8])"
}
export GM_SMALL_TABLE_MIN=11
GM_SMALL_TABLE_C=0 GM_FLAT_TABLE_MAX=19 run
GM_SMALL_TABLE_C=16 GM_FLAT_TABLE_MAX=19 run
GM_SMALL_TABLE_C=16 GM_FLAT_TABLE_MAX=17 run
GM_SMALL_TABLE_C=16 GM_FLAT_TABLE_MAX=0 run
GM_SMALL_TABLE_C=18 GM_FLAT_TABLE_MAX=19 run
GM_SMALL_TABLE_C=17 GM_FLAT_TABLE_MAX=19 run
GM_SMALL_TABLE_C=0 GM_FLAT_TABLE_MAX=19 run
