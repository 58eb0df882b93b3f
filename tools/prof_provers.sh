#!/bin/bash
# GPU box: kernel trace + stats of the two larger provers (psnark -i 22, elastic snark -i 24).  Usage: tools/prof_provers.sh <tag>
TAG=${1:-r2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/psnark22 -o psnark22 -- python $GRAFT_REPO_ROOT/tools/run_psnark.py -i 22 --repeat 2 > $OUT/psnark22_run.json 2> $OUT/psnark22.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/elastic24 -o elastic24 -- python $GRAFT_REPO_ROOT/tools/run_snark.py -i 24 --elastic --repeat 2 > $OUT/elastic24_run.json 2> $OUT/elastic24.err
find $OUT -name "*kernel_trace.csv" -size +8M -delete
for d in psnark22 elastic24; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); echo "== $d"; head -25 $f | cut -c1-160; done
tail -3 $OUT/psnark22.err $OUT/elastic24.err
