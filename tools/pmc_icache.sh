#!/bin/bash
# instruction-cache and issue counters of k_acc0 (one --pmc pass per group; --pmc only with --kernel-trace)
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_icache
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST_ANY\|SQ_WAIT_ANY\|SQ_BUSY_CYCLES\|SQ_WAVE_CYCLES\|SQ_INSTS_VALU\b\|SQ_ACTIVE_INST_VALU\|SQ_INST_CYCLES_[A-Z]*\|SQ_ACTIVE_INST_ANY\|SQ_WAIT_INST_LDS\|SQ_INSTS_SALU\|SQ_THREAD_CYCLES_VALU\|GRBM_GUI_ACTIVE" | sort -u > $OUT/avail.txt
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_IFETCH SQ_WAIT_ANY"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$tag -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --headline-only > /dev/null 2> $OUT/$tag.err
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  python - "$f" <<PY
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if "k_acc0" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items()):
    print(f"k_acc0 {k}: {v / n:.4g} per launch over {n} launches")
PY
done 2>&1 | tee $OUT/summary.txt
find $OUT -name "*.csv" -size +4M -delete
