#!/bin/bash
# per-kernel times of tools/run_snark.py -i <logn> (SRS generation included, listed separately by name)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_snark -o s -- python $GRAFT_REPO_ROOT/tools/run_snark.py -i $1 --repeat 3 > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/prof_snark/s_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)/1e6
print("total kernel ms (3 proofs + setup): %.1f"%tot)
for r in rows[:28]:
    print("%-44s calls %5s avg_us %10.1f total_ms %9.2f"%(r["Name"].split("(")[0][:44], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
