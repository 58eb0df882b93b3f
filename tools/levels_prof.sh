#!/bin/bash
# per-kernel times of one affine-level configuration: tools/levels_prof.sh <logn> <levels>
cd /tmp && export TMPDIR=/tmp
GM_PROBE_LEVELS=$2 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_lvl -o lvl -- python $GRAFT_REPO_ROOT/tools/levels_probe.py $1 > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/prof_lvl/lvl_kernel_stats.csv")))
for r in rows[:16]:
    print("%-60s calls %5s avg_us %10.1f total_ms %9.2f"%(r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
