#!/bin/bash
# HBM traffic counters of the field kernels of a `snark -i 24` proof (one --pmc pass each, --kernel-trace only): per-kernel sums
# over ONE run of two proofs, printed per proof.  Usage (GPU box): tools/pmc_snark.sh > gpurun_out/pmc_snark24.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_snark24
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o pmc -- python $GRAFT_REPO_ROOT/tools/run_snark.py -i 24 --repeat 2 --native > /dev/null 2> $OUT/$c.err
done
python - <<PY
import csv, glob, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True)[0]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        a = acc[n]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, (n, v) in acc.items():
        out.setdefault(k, {})["dispatches"] = n
        out[k][c + "_KiB_total"] = round(v, 1)
rows = sorted(out.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE_KiB_total", 0) + kv[1].get("WRITE_SIZE_KiB_total", 0)))
print(json.dumps({"command": "tools/run_snark.py -i 24 --repeat 2 --native (two proofs + setup)", "unit": "KiB summed over all dispatches of the run, uncorrected",
                  "kernels": dict(rows[:25])}, indent=1))
PY
find $OUT -name "*.csv" -size +4M -delete
