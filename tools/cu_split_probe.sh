#!/bin/bash
# GPU box: the CU partition of a batch (GM_CU_SPLIT=T compute units for the tails) against the default, same box, same process order
O=${1:-gpurun_out/r5_cu_split_probe.txt}
: > $O
run() {  # label env... -- command
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['runs']
k=[x for x in r[0] if 'prover' in x][0]
print('$label', ' '.join('${envs[*]}'.split()), 'best', min(x[k] for x in r), 'median', sorted(x[k] for x in r)[len(r)//2], 'tensorcheck', min(x.get('Tensorcheck',0) for x in r), d['proof_sha256'][:8])" >> $O
}
for T in 0 16 32 64; do
  for S in acc tail; do
    [ $T = 0 ] && [ $S = tail ] && continue
    run shard21 GM_CU_SPLIT=$T GM_CU_SPLIT_SORT=$S -- python tools/run_snark.py -i 21 --repeat 7 --block-sharded --transport shm
    run snark24 GM_CU_SPLIT=$T GM_CU_SPLIT_SORT=$S -- python tools/run_snark.py -i 24 --repeat 4 --native
    run snark20 GM_CU_SPLIT=$T GM_CU_SPLIT_SORT=$S -- python tools/run_snark.py -i 20 --repeat 7 --native
    run psnark22 GM_CU_SPLIT=$T GM_CU_SPLIT_SORT=$S -- python tools/run_psnark.py -i 22 --repeat 3 --native
  done
done
run shard21_smalltail GM_CU_SPLIT=32 GM_CU_SPLIT_SMALL=tail -- python tools/run_snark.py -i 21 --repeat 7 --block-sharded --transport shm
run psnark22_smalltail GM_CU_SPLIT=32 GM_CU_SPLIT_SMALL=tail -- python tools/run_psnark.py -i 22 --repeat 3 --native
cat $O
