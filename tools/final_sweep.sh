#!/bin/bash
# GPU box: the prover sweep of a round's final library (best of the repeats; every prover through its compiled entry point)
O=gpurun_out/final_sweep.txt
: > $O
one() {  # label, command...
  local label=$1; shift
  "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$label', d.get('time_prover_s', d.get('elastic_prover_s')), d['proof_sha256'][:8])" >> $O
}
if [ -z "$ELASTIC_ONLY" ]; then
for i in 18 20 22 24 26 27 28; do one "snark_time_$i" python tools/run_snark.py -i $i --repeat 3 --native; done
for i in 18 20 22 24; do one "psnark_time_$i" python tools/run_psnark.py -i $i --repeat 2 --native; done
one "sharded_world1_21" python tools/run_snark.py -i 21 --repeat 5 --block-sharded --transport shm
fi
for i in 22 24 26 28; do one "snark_elastic_$i" python tools/run_snark.py -i $i --repeat 2 --elastic --dummy-srs --native; done
for i in 20 22 24; do one "psnark_elastic_$i" python tools/run_psnark.py -i $i --repeat 2 --elastic --native; done
cat $O
