#!/bin/bash
# GPU box: what one rank of g GPUs does for `psnark -i $TOP` with every vector block-sharded (gm_psnark_new_time_sharded), timed on ONE GPU as the
# same prover with world = 1 at -i ($TOP - log2 g).  world 1 at -i (TOP - log2 g) processes 1 / g of EVERY vector: the AVERAGE rank's share.  With one
# block size for all vectors the ranks below g / 2 hold a block of all 22 base polynomials, the ranks above only of the 6 that are ~2 n long: the
# busiest rank carries ~46 / 33 of the average MSM work, which the last column prices in.  On top come the whole-vector passes
# every rank repeats (tensor(rho), powers(alpha), the hashed sets: O(n) at HBM speed, timed here as the world-1 spans at the TOP size would be) and
# the collectives.  Falsifiable by the first SCALE run.
TOP=${1:-26}
O=${2:-gpurun_out/r6_psnark_shard_shares.txt}
one() { "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); k='time_prover_s'; r=sorted(x['ark_gemini::psnark::time_prover'] for x in d['runs']); print(min(r), r[len(r)//2], d['proof_sha256'][:8], d['mem_GB']['in_use_peak'])"; }
echo "# one MI355X, $(date -u +%F), library $(sha256sum gemini_amd/libgemini_hip.so | cut -c1-12); seconds: best, median; proof hash; peak GB in use" > $O
read b1 m1 h1 g1 <<< $(one python tools/run_psnark.py -i $TOP --repeat 3 --native)
echo "g=1  psnark -i $TOP, one GPU (gm_psnark_new_time)                      $b1 $m1 $h1 ${g1} GB" >> $O
for g in 2 4 8; do
  lg=$(python -c "import math; print($TOP - int(math.log2($g)))")
  read b m h gb <<< $(one python tools/run_psnark.py -i $lg --repeat 5 --block-sharded --transport shm)
  echo "g=$g  average rank's share = world 1 at -i $lg (gm_psnark_new_time_sharded)  $b $m $h ${gb} GB   speed-up if balanced: $(python -c "print(round($b1 / $b, 2))") x (efficiency $(python -c "print(round($b1 / $b / $g, 2))")); busiest rank at 46/33 of the MSM share (~75 % of the time): $(python -c "print(round($b1 / ($b * (0.25 + 0.75 * 46 / 33)), 2))") x" >> $O
done
cat $O
