#!/bin/bash
# GPU box: per-kernel rocprofv3 stats of a command under several settings of one environment variable.
# Usage: tools/kstats_probe.sh VAR "v0 v1" <kernel-name regex> <command...>
VAR=$1; VALS=$2; PAT=$3; shift 3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in $VALS; do
  export $VAR=$v
  O=$R/gpurun_out/kstats_${VAR}_$v
  rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- "$@" > $O/run.json 2> $O/err.txt
  echo "== $VAR=$v"
  python - "$O" "$PAT" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if re.search(sys.argv[2], r["Name"]):
        print("%-28s calls %5s total %9.3f ms avg %9.2f us max %9.2f us" % (r["Name"].split("(")[0][-28:], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  find $O -name "*kernel_trace.csv" -delete
done
