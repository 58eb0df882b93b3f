#!/bin/bash
# GPU box: kernel trace of the folding-commitment batch (tools/batch_probe.py) under rocprofv3 for settings "T[:small]" of the CU
# partition (GM_CU_SPLIT=T, GM_CU_SPLIT_SMALL=tail with ":small"); the traces stay in gpurun_out/ for tools/batch_timeline.py
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for spec in ${@:-0 32}; do
  T=${spec%%:*}; SM=""; [[ $spec == *:small ]] && SM=tail
  O=$R/gpurun_out/batch_tl_${spec/:/_}
  rm -rf $O; mkdir -p $O
  GM_CU_SPLIT=$T GM_CU_SPLIT_SMALL=$SM PROBE_BATCH_ONLY=1 rocprofv3 --kernel-trace --output-format csv -d $O -o k -- python $R/tools/batch_probe.py 20 tables > $O/run.txt 2> $O/err.txt
  echo "== GM_CU_SPLIT=$T small=$SM: $(tail -1 $O/run.txt)"
done
