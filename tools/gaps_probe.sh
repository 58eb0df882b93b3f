#!/bin/bash
# GPU box: exposed-time table + the largest GPU-idle gaps of the last proof of a prover run.  Usage: tools/gaps_probe.sh <name> <N gaps> <command...>
NAME=$1; N=$2; shift 2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/gaps_$NAME
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- "$@" > $O/run.json 2> $O/err.txt
tr=$(find $O/t -name "*kernel_trace.csv" | head -1)
python $R/tools/exposed_time.py $tr --stamps $O/run.json --gaps $N --title "$NAME: $*" --md $O/exposed.md > $O/gaps.txt
rm -f $tr
cat $O/gaps.txt
