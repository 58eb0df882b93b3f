#!/bin/bash
# GPU box: `snark` elastic prover on the generator-copies key over INSTANCE_LOGSIZE (examples/snark.rs:54-66).  Usage: tools/elastic_sweep.sh out.jsonl [sizes]
OUT=${1:-gpurun_out/elastic_sweep.jsonl}
: > $OUT
for i in ${SIZES:-22 24 26 28}; do
  timeout 900 python tools/run_snark.py -i $i --repeat 2 --elastic --dummy-srs >> $OUT 2> /dev/null || echo "{\"logn\": $i, \"error\": \"rc=$?\"}" >> $OUT
done
python - <<PY
import json
for line in open("$OUT"):
    d = json.loads(line)
    print(d.get("logn"), d.get("elastic_prover_s"), d.get("proof_sha256", "")[:16], d.get("error", ""))
PY
