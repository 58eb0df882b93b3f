#!/bin/bash
# per-kernel times of one-call MSMs at the given log sizes: tools/sort_prof.sh 20 24
cd /tmp && export TMPDIR=/tmp
GM_PROBE_LEVELS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_sort -o s -- python $GRAFT_REPO_ROOT/tools/levels_probe.py "$@" > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/prof_sort/s_kernel_stats.csv")))
for r in rows[:14]:
    print("%-44s calls %5s avg_us %10.1f max_us %10.1f total_ms %9.2f"%(r["Name"].split("(")[0][:44], r["Calls"], float(r["AverageNs"])/1e3, float(r["MaxNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
