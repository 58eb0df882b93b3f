#!/bin/bash
# GPU box: what one rank of g GPUs does for `snark -i 24` with every vector block-sharded, timed on ONE GPU as the same prover
# with world = 1 at -i (24 - log2 g): g = 2, 4, 8 -> -i 23, 22, 21 (the collectives come on top: 25 per proof, <= 0.5 ms over the
# node's segment, + the re-blocking of the opening: (log2 g + 1) m elements into rank 0, ~1 ms of one xGMI link at g = 8).
# The prediction is falsifiable by the first SCALE run: time(N) ~ share(N) + collectives.
O=${1:-gpurun_out/r5_shard_shares.txt}
one() { "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=sorted(x['ark_gemini::snark::time_prover'] for x in d['runs']); print(min(r), r[len(r)//2], d['proof_sha256'][:8])"; }
echo "# one MI355X, $(date -u +%F), library $(sha256sum gemini_amd/libgemini_hip.so | cut -c1-12); seconds: best, median of 7; proof hash" > $O
read b1 m1 h1 <<< $(one python tools/run_snark.py -i 24 --repeat 5 --native)
echo "g=1  snark -i 24, one GPU (gm_snark_new_time)            $b1 $m1 $h1" >> $O
for g in 2 4 8; do
  lg=$(python -c "import math; print(24 - int(math.log2($g)))")
  read b m h <<< $(one python tools/run_snark.py -i $lg --repeat 7 --block-sharded --transport shm)
  sp=$(python -c "print(round($b1 / $b, 2))")
  echo "g=$g  one rank's share = world 1 at -i $lg (gm_snark_new_time_sharded)  $b $m $h   predicted speed-up before collectives: $sp x (efficiency $(python -c "print(round($b1 / $b / $g, 2))"))" >> $O
done
GM_SHARD_TRACE=1 python tools/run_snark.py -i 21 --repeat 3 --block-sharded --transport shm 2>&1 >/dev/null | grep "gm shard" | tail -7 >> $O
cat $O
