#!/bin/bash
# GPU box: what one rank of g GPUs does for `psnark -i $TOP` with every vector block-sharded (gm_psnark_new_time_sharded), timed on ONE GPU as the
# same prover with world = 1 at -i ($TOP - log2 g): it processes 1 / g of EVERY vector, which with per-family levels (gm_psnark_shard_level: every rank
# holds ~1 / g of every vector, short or long) is what each rank does, re-blocking of the cross-level combinations included (a device copy at world 1,
# <= ~6 n / g elements over xGMI on a node).  On top come the whole-vector passes every rank repeats at the TOP size (tensor(rho), powers(alpha), the
# hashed sets: O(n) at HBM speed, ~2 ms each at 2^26) and the collectives.  Falsifiable by the first SCALE run.
TOP=${1:-26}
O=${2:-gpurun_out/r6_psnark_shard_shares.txt}
one() { "$@" 2>/dev/null | tee -a ${O}.raw | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); k='time_prover_s'; r=sorted(x['ark_gemini::psnark::time_prover'] for x in d['runs']); print(min(r), r[len(r)//2], d['proof_sha256'][:8], d['mem_GB']['in_use_peak']); sys.stderr.write(json.dumps({'logn': d['logn'], 'mem_GB': d['mem_GB'], 'layout': d.get('layout'), 'spans': d['runs'][-1]}) + chr(10))"; }
echo "# one MI355X, $(date -u +%F), library $(sha256sum gemini_amd/libgemini_hip.so | cut -c1-12); seconds: best, median; proof hash; peak GB in use" > $O
read b1 m1 h1 g1 <<< $(one python tools/run_psnark.py -i $TOP --repeat 3 --native)
echo "g=1  psnark -i $TOP, one GPU (gm_psnark_new_time)                      $b1 $m1 $h1 ${g1} GB" >> $O
for g in 2 4 8; do
  lg=$(python -c "import math; print($TOP - int(math.log2($g)))")
  read b m h gb <<< $(one python tools/run_psnark.py -i $lg --repeat 5 --block-sharded --transport shm)
  echo "g=$g  one rank's share = world 1 at -i $lg (gm_psnark_new_time_sharded)  $b $m $h ${gb} GB   predicted speed-up before collectives: $(python -c "print(round($b1 / $b, 2))") x (efficiency $(python -c "print(round($b1 / $b / $g, 2))"))" >> $O
done
cat $O
