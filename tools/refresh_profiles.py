#!/usr/bin/env python3
"""Copy the measurement artefacts of a `gpurun` session from gpurun_out/ into profiles/ (tracked):
bench JSON, rocprofv3 kernel stats (csv + markdown), PMC traffic per kernel, prover / size sweeps."""
import collections
import csv
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")

rows = list(csv.DictReader(open(os.path.join(G, "prof_final", "msm20_kernel_stats.csv"))))
out = ["# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --headline-only (MI355X, round 1 final library)", "",
       "23 one-call 2^20-pair MSMs (3 warm-up + 20 timed), nothing else in the command.", "",
       "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
for r in rows:
    out.append("| `%s` | %s | %.3f | %.2f | %.2f | %.2f | %s |" % (r["Name"].split("(")[0], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                                                              float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
open(os.path.join(P, "r1_final_msm20_kernel_stats.md"), "w").write("\n".join(out) + "\n")
shutil.copy(os.path.join(G, "prof_final", "msm20_kernel_stats.csv"), os.path.join(P, "r1_final_msm20_kernel_stats.csv"))
shutil.copy(os.path.join(G, "bench_final.json"), os.path.join(P, "r1_final_bench.json"))

res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(G, f"pmc_{c}", "pmc_counter_collection.csv"))):
        if r["Counter_Name"] == c:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    res[c] = {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}
old = json.load(open(os.path.join(P, "r1_pmc_msm20.json")))
pm = {"source": old["source"], "units": old["units"], "kernels": {}}
for k, (f, n) in sorted(res["FETCH_SIZE"].items()):
    if "gm::" not in k:
        continue
    w = res["WRITE_SIZE"].get(k, (0, 0))[0]
    pm["kernels"][k] = {"dispatches": n, "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "hbm_bytes_uncorrected": int((f + w) * 1024),
                        "hbm_bytes_corrected": int((2 * f + w) * 1024)}
json.dump(pm, open(os.path.join(P, "r1_pmc_msm20.json"), "w"), indent=1)
for src, dst in (("time_prover_sweep.jsonl", "r1_time_prover_sweep.jsonl"), ("psnark_sweep.jsonl", "r1_psnark_time_prover_sweep.jsonl"),
                 ("msm_sizes_final.txt", "r1_final_msm_sizes.txt")):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, dst))
print("k_acc0 PMC:", pm["kernels"].get("gm::k_acc0"))
print("k_acc0 rocprof avg us:", [float(r["AverageNs"]) / 1e3 for r in rows if r["Name"].startswith("gm::k_acc0")])
