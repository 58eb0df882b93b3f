#!/bin/bash
# GPU box, round 5: the prover sweep with the library's memory bookkeeping beside every time (gm_mem_stats: high-water mark of the
# bytes IN USE during the proofs, tables and keys apart).  One process per line so that a failure at a large size loses nothing else.
O=${1:-gpurun_out/${TAG:-r6}_prover_sweep.txt}
: > $O
one() {  # label, command...
  local label=$1; shift
  timeout ${PER_RUN_TIMEOUT:-900} "$@" 2>gpurun_out/${TAG:-r6}_err_$label.log | python -c "
import sys,json
L=sys.stdin.readlines()
try:
    d=json.loads(L[-1]); m=d.get('mem_GB',{})
    print('$label', d.get('time_prover_s', d.get('elastic_prover_s')), 's', d['proof_sha256'][:8], 'peak_in_use_GB', m.get('in_use_peak'), 'held_peak_GB', m.get('held_peak'), 'tables_GB', m.get('tables'), 'keys_GB', m.get('keys'), 'spare_releases', d.get('spare_table_releases'))
except Exception as e:
    print('$label', 'FAILED', repr(e))
" >> $O
  tail -3 gpurun_out/${TAG:-r6}_err_$label.log | grep -i "error\|ENOMEM\|Traceback" >> $O
}
for i in ${PSNARK_SIZES:-20 22 24 26}; do
  one "psnark_elastic_$i" python tools/run_psnark.py -i $i --repeat ${REPEAT:-2} --elastic --native
  one "psnark_time_$i" python tools/run_psnark.py -i $i --repeat ${REPEAT:-2} --native
done
for i in ${SNARK_SIZES:-20 22 24 26 28}; do
  one "snark_time_$i" python tools/run_snark.py -i $i --repeat ${REPEAT:-2} --native
  one "snark_elastic_$i" python tools/run_snark.py -i $i --repeat ${REPEAT:-2} --elastic --dummy-srs --native
done
cat $O
