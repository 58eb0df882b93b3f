#!/usr/bin/env python3
"""Copy the measurement artefacts of tools/profile_round.sh <tag> from gpurun_out/prof_<tag>/ into profiles/ (tracked):
rocprofv3 kernel stats of the headline command and of `snark -i 24` (csv + markdown) and the PMC traffic per kernel.
Usage: python tools/refresh_profiles.py r2 "library description" """
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r2"
DESC = sys.argv[2] if len(sys.argv) > 2 else "round 2 library"
G = os.path.join(ROOT, "gpurun_out", f"prof_{TAG}")
P = os.path.join(ROOT, "profiles")


def kname(full):
    """`void gm::k_acc0<2>(unsigned long const*, ...)` -> `gm::k_acc0`"""
    n = full.split("(")[0].strip()
    if n.startswith("void "):
        n = n[5:]
    return n.split("<")[0]


def find(sub, pattern):
    hits = glob.glob(os.path.join(G, sub, "**", pattern), recursive=True)
    assert hits, (sub, pattern)
    return hits[0]


def stats_md(csv_path, title, note, out_name, top=40):
    rows = list(csv.DictReader(open(csv_path)))
    out = [f"# {title}", "", note, "", "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
    for r in rows[:top]:
        out.append("| `%s` | %s | %.3f | %.2f | %.2f | %.2f | %s |" % (kname(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                                                                  float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
    open(os.path.join(P, out_name + ".md"), "w").write("\n".join(out) + "\n")
    shutil.copy(csv_path, os.path.join(P, out_name + ".csv"))
    return rows


rows = stats_md(find("msm20", "*kernel_stats.csv"),
                f"rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --headline-only (MI355X, {DESC})",
                "33 one-call 2^20-pair MSMs (3 warm-up + 20 timed + 10 untimed ones with every stage timer on), nothing else in the command.", f"{TAG}_msm20_kernel_stats")
shutil.copy(os.path.join(G, "msm20_bench.json"), os.path.join(P, f"{TAG}_msm20_bench_under_rocprof.json"))
stats_md(find("snark24", "*kernel_stats.csv"),
         f"rocprofv3 --kernel-trace --stats -- python tools/run_snark.py -i 24 --repeat 3 (MI355X, {DESC})",
         "dummy_r1cs(2^24) + SRS generation (2^25 + 1 points, listed by their own kernel names) + three Proof::new_time runs.", f"{TAG}_snark24_kernel_stats")
shutil.copy(os.path.join(G, "snark24_run.json"), os.path.join(P, f"{TAG}_snark24_run.json"))

res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(find(f"pmc_{c}", "*counter_collection.csv"))):
        if r["Counter_Name"] == c:
            acc[kname(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    res[c] = {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}
pm = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on `python bench.py --steps 3 --warmup 1 --headline-only`, "
                f"MI355X, {DESC}; averages over all dispatches of a kernel in that command (seven one-call 2^20-pair MSMs: 1 warm-up + 3 timed + 3 with every stage timer on)",
      "units": "counter values are KiB per dispatch; bytes = value * 1024; gfx950 correction per MI355X_MICROARCH.md section HBM: FETCH_SIZE tallies 128-B requests "
               "at 64 B for 16-B/lane loads, so read bytes = 2 * FETCH_SIZE * 1024 (calibrated there on streaming reads, uncalibrated for this gather pattern; "
               "Infinity-Cache hits are included); WRITE_SIZE as reported",
      "kernels": {}}
for k, (f, n) in sorted(res["FETCH_SIZE"].items()):
    if "gm::" not in k:
        continue
    w = res["WRITE_SIZE"].get(k, (0, 0))[0]
    pm["kernels"][k] = {"dispatches": n, "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "hbm_bytes_uncorrected": int((f + w) * 1024),
                        "hbm_bytes_corrected": int((2 * f + w) * 1024)}
json.dump(pm, open(os.path.join(P, f"{TAG}_pmc_msm20.json"), "w"), indent=1)
print("k_acc0 PMC:", pm["kernels"].get("gm::k_acc0"))
print("k_acc0 rocprof avg us:", [float(r["AverageNs"]) / 1e3 for r in rows if kname(r["Name"]) == "gm::k_acc0"])
