#!/bin/bash
# GPU box: kernel traces of the three provers + the exposed-time tables (tools/exposed_time.py) of their LAST proof.
# Usage: tools/exposed_round.sh <tag>   -> gpurun_out/exposed_<tag>/{*.md,*.json}; copy the .md / .json into profiles/.
TAG=${1:-r3}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/exposed_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # name, command...
  local name=$1; shift
  rocprofv3 --kernel-trace --output-format csv -d $O/$name -o $name -- "$@" > $O/${name}_run.json 2> $O/$name.err
  local tr=$(find $O/$name -name "*kernel_trace.csv" | head -1)
  python $R/tools/exposed_time.py $tr --stamps $O/${name}_run.json --title "$name: $*" --md $O/exposed_$TAG.md --json $O/${name}_exposed.json > /dev/null
  rm -f $tr   # the per-dispatch trace is large; the table is what is kept
}
: > $O/exposed_$TAG.md
run snark24 python $R/tools/run_snark.py -i 24 --repeat 3 --native
run elastic24 python $R/tools/run_snark.py -i 24 --repeat 3 --elastic --dummy-srs --native
run psnark22 python $R/tools/run_psnark.py -i 22 --repeat 3 --native
run psnark22_elastic python $R/tools/run_psnark.py -i 22 --repeat 3 --elastic --native
run shard21 python $R/tools/run_snark.py -i 21 --repeat 3 --block-sharded --transport shm
run pshard19 python $R/tools/run_psnark.py -i 19 --repeat 3 --block-sharded --transport shm
cat $O/exposed_$TAG.md
