#!/bin/bash
# GPU box: the rocprofv3 evidence of a round.  Usage: tools/profile_round.sh <tag>      (e.g. r2)
#   1. kernel trace + stats of the headline command (one-call 2^20 MSMs only)
#   2. HBM traffic counters, one --pmc pass each (FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc is never
#      combined with anything but --kernel-trace)
#   3. kernel trace + stats of three `snark --time-prover -i 24` proofs
# Everything lands under gpurun_out/prof_<tag>/; tools/refresh_profiles.py <tag> copies the summaries to profiles/.
TAG=${1:-r2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/msm20 -o msm20 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --headline-only > $OUT/msm20_bench.json 2> $OUT/msm20.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --headline-only > /dev/null 2> $OUT/pmc_$c.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/snark24 -o snark24 -- python $GRAFT_REPO_ROOT/tools/run_snark.py -i 24 --repeat 3 --native > $OUT/snark24_run.json 2> $OUT/snark24.err
find $OUT -name "*.csv" | head -20
# keep the merge-back small: the per-dispatch traces are large, the stats and counter tables are what is read
find $OUT -name "*kernel_trace.csv" -size +8M -delete
du -sh $OUT
