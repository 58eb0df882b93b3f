#!/usr/bin/env python3
"""Per-queue timeline of the LAST gm_g1_msm_v_batch of tools/batch_probe.py from a rocprofv3 kernel trace (dev tool):
which queue ran what when, how many kernels were in flight over time."""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        if r.get("Kind", "KERNEL_DISPATCH") != "KERNEL_DISPATCH":
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gm::", "").split("<")[0], r.get("Queue_Id", "?")))
rows.sort()
# the probe ends with per-level single calls (3 x 20 + 2 x 20 = 100 MSMs) after 5 timed batches: find the batches as windows between large idle gaps
gaps = []
end = rows[0][1]
for i, (s, e, n, q) in enumerate(rows[1:], 1):
    if s - end > 10_000_000:
        gaps.append(i)
    end = max(end, e)
wins = [rows[a:b] for a, b in zip([0] + gaps, gaps + [len(rows)])]
# (a batch of 20 levels is 20 accumulations, or 9 with the tiny calls fused into one pass: MsmMulti)
batches = [w for w in wins if 8 <= sum(1 for x in w if x[2] in ("k_acc0", "k_acc0_pf")) <= 20 and sum(1 for x in w if x[2].startswith("k_sort1")) <= 20]
w = batches[int(sys.argv[2]) if len(sys.argv) > 2 else -1]
t0 = w[0][0]
print("window %.3f ms, %d kernels, queues %s" % ((max(x[1] for x in w) - t0) / 1e6, len(w), sorted({x[3] for x in w})))
byq = defaultdict(list)
for s, e, n, q in w:
    byq[q].append((s, e, n))
for q in sorted(byq):
    ev = byq[q]
    busy = sum(e - s for s, e, _ in ev)
    print("queue %s: %d kernels, busy %.3f ms, first %.3f last %.3f" % (q, len(ev), busy / 1e6, (ev[0][0] - t0) / 1e6, (ev[-1][1] - t0) / 1e6))
# concurrency histogram
pts = []
for s, e, n, q in w:
    pts.append((s, 1))
    pts.append((e, -1))
pts.sort()
cur, last, hist = 0, t0, defaultdict(int)
for t, d in pts:
    hist[cur] += t - last
    cur += d
    last = t
print("time with k kernels in flight (ms):", {k: round(v / 1e6, 3) for k, v in sorted(hist.items())})
for s, e, n, q in w:
    if len(sys.argv) > 3:
        print("%8.3f %8.3f %-20s q%s" % ((s - t0) / 1e6, (e - s) / 1e6, n, q))
