#!/bin/bash
# GPU box: the measurement batch of a round (after tools/profile_round.sh): size sweep of one-call MSMs, the
# time-prover / elastic / preprocessing prover sweeps over INSTANCE_LOGSIZE, instruction-rate microbenchmarks, the
# full bench line.  Usage: tools/measure_round.sh <tag>; tools/refresh_profiles.py copies nothing from here --
# the files are small and are copied to profiles/<tag>_* by hand (see profiles/README.md).
TAG=${1:-r2}
O=gpurun_out/measure_$TAG
mkdir -p $O
python tools/msm_sizes.py > $O/msm_sizes.txt 2> $O/msm_sizes.err
SIZES="20 22 24 26 27 28" tools/sweep_snark.sh $O/time_prover_sweep.jsonl > /dev/null 2>&1
: > $O/elastic_sweep.jsonl
for i in 22 24 26 28; do
  timeout 900 python tools/run_snark.py -i $i --repeat 2 --elastic --dummy-srs >> $O/elastic_sweep.jsonl 2> $O/elastic_err_$i.log || echo "{\"logn\": $i, \"error\": \"rc=$?\"}" >> $O/elastic_sweep.jsonl
done
: > $O/psnark_sweep.jsonl
for i in 20 22 24 26; do
  timeout 900 python tools/run_psnark.py -i $i --repeat 2 >> $O/psnark_sweep.jsonl 2> $O/psnark_err_$i.log || echo "{\"logn\": $i, \"error\": \"rc=$?\"}" >> $O/psnark_sweep.jsonl
done
tools/_build/fqmul_check > $O/fqmul_check.txt 2>&1
tools/_build/ubench_regs > $O/ubench_regs.txt 2>&1
python tools/degenerate_probe.py 20 24 > $O/degenerate_probe.txt 2>&1
python tools/small_probe.py > $O/small_probe.txt 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
ls -la $O
