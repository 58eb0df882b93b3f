#!/bin/bash
# GPU box: kernel timeline of the folding-commitment batch (tools/batch_probe.py) under rocprofv3, for each GM_CU_SPLIT value given
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for T in ${@:-0 32}; do
  O=$R/gpurun_out/batch_tl_$T
  rm -rf $O; mkdir -p $O
  GM_CU_SPLIT=$T PROBE_BATCH_ONLY=1 rocprofv3 --kernel-trace --output-format csv -d $O -o k -- python $R/tools/batch_probe.py 20 tables > $O/run.txt 2> $O/err.txt
  echo "== GM_CU_SPLIT=$T"; tail -3 $O/run.txt
  python $R/tools/batch_timeline.py $(find $O -name "*kernel_trace.csv" | head -1) -1 full > $R/gpurun_out/r5_batch_timeline_split$T.txt 2>&1
  head -60 $R/gpurun_out/r5_batch_timeline_split$T.txt

done
