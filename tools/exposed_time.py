#!/usr/bin/env python3
"""Turn a rocprofv3 kernel trace into an EXPOSED-TIME table.

Per-kernel sums (what `--stats` prints) say nothing about wall time once two streams overlap: the sum of
`k_acc0` is stretched by every kernel that shares the CUs with it, and a latency-bound tail kernel that runs
in the shadow of an accumulation costs nothing.  This tool answers the question the prover timings pose:

    wall = (time some k_acc0 is running)  +  (time only OTHER kernels are running, by kernel)  +  (GPU idle)

over one window of the trace -- by default the LAST repetition of the run (the trace is cut at the largest
`--reps - 1` idle gaps; a run script that prints clock stamps can pass `--window t0,t1` in trace nanoseconds
instead).  A moment covered by several non-primary kernels is split evenly between them.

Usage: tools/exposed_time.py TRACE.csv [--primary k_acc0] [--reps 3] [--window t0,t1] [--md out.md] [--json out.json]
"""
import argparse
import csv
import json
import re
import sys
from collections import defaultdict


def short_name(full):
    m = re.match(r"(?:void\s+)?(?:gm::)?([A-Za-z_0-9]+)", full)
    n = m.group(1) if m else full
    return "k_acc0" if n == "k_acc0_pf" else n  # the same accumulation with the next gather in flight (table path)


def load(path):
    ev = []
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row.get("Kind", "KERNEL_DISPATCH") != "KERNEL_DISPATCH":
                continue
            ev.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), short_name(row["Kernel_Name"])))
    ev.sort()
    return ev


def cut_last_rep(ev, reps):
    """window of the last repetition: cut at the (reps - 1) largest idle gaps of the whole trace"""
    if reps <= 1 or len(ev) < 2:
        return ev[0][0], max(e[1] for e in ev)
    gaps = []
    cover_end = ev[0][1]
    for s, e, _ in ev[1:]:
        if s > cover_end:
            gaps.append((s - cover_end, cover_end, s))
        cover_end = max(cover_end, e)
    gaps.sort(reverse=True)
    cuts = sorted(g[2] for g in gaps[: reps - 1])
    t0 = cuts[-1] if cuts else ev[0][0]
    return t0, max(e[1] for e in ev)


def analyse(ev, t0, t1, primary):
    ev = [(max(s, t0), min(e, t1), n) for s, e, n in ev if e > t0 and s < t1]
    pts = []
    for s, e, n in ev:
        pts.append((s, 1, n))
        pts.append((e, -1, n))
    pts.sort(key=lambda p: (p[0], p[1]))
    live = defaultdict(int)
    exposed = defaultdict(float)
    busy_sum = defaultdict(float)
    count = defaultdict(int)
    for s, e, n in ev:
        busy_sum[n] += e - s
        count[n] += 1
    prim_t = 0.0
    idle = 0.0
    prev = t0
    for t, d, n in pts:
        dt = t - prev
        if dt > 0:
            names = [k for k, v in live.items() if v > 0]
            if any(k == primary for k in names):
                prim_t += dt
            elif names:
                for k in names:
                    exposed[k] += dt / len(names)
            else:
                idle += dt
        prev = t
        live[n] += d
    idle += max(0, t1 - prev)
    wall = t1 - t0
    rows = sorted(((k, v) for k, v in exposed.items()), key=lambda kv: -kv[1])
    return {
        "wall_ms": wall / 1e6,
        "primary": primary,
        "primary_ms": prim_t / 1e6,
        "primary_share": prim_t / wall if wall else 0.0,
        "other_exposed_ms": sum(v for _, v in rows) / 1e6,
        "idle_ms": idle / 1e6,
        "idle_share": idle / wall if wall else 0.0,
        "exposed_by_kernel_ms": {k: v / 1e6 for k, v in rows},
        "kernel_sum_ms": {k: v / 1e6 for k, v in sorted(busy_sum.items(), key=lambda kv: -kv[1])},
        "launches": dict(count),
    }


def to_md(r, title):
    out = [f"### {title}", "",
           f"wall {r['wall_ms']:.2f} ms = `{r['primary']}` running {r['primary_ms']:.2f} ({100 * r['primary_share']:.1f} %)"
           f" + other kernels exposed {r['other_exposed_ms']:.2f} + GPU idle {r['idle_ms']:.2f} ({100 * r['idle_share']:.1f} %)", "",
           "| kernel | launches | exposed ms (not under the primary) | sum of durations ms |", "|---|---|---|---|"]
    names = list(r["exposed_by_kernel_ms"].keys())
    for k in r["kernel_sum_ms"]:
        if k not in names:
            names.append(k)
    for k in names:
        ex = r["exposed_by_kernel_ms"].get(k, 0.0)
        if ex < 0.005 and r["kernel_sum_ms"].get(k, 0.0) < 0.05:
            continue
        out.append(f"| `{k}` | {r['launches'].get(k, 0)} | {ex:.2f} | {r['kernel_sum_ms'].get(k, 0.0):.2f} |")
    return "\n".join(out) + "\n"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--primary", default="k_acc0")
    ap.add_argument("--reps", type=int, default=1, help="repetitions in the run; the window is the last one")
    ap.add_argument("--window", default=None, help="t0,t1 in trace nanoseconds")
    ap.add_argument("--stamps", default=None, help="JSON output of tools/run_snark.py / run_psnark.py of the traced run: the window is its LAST proof "
                    "(clock readings around every proof; the clock that brackets kernel activity is picked)")
    ap.add_argument("--gaps", type=int, default=0, help="also list the N largest GPU-idle gaps of the window with the kernels around them")
    ap.add_argument("--title", default=None)
    ap.add_argument("--md", default=None)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    ev = load(a.trace)
    if not ev:
        sys.exit("no kernel dispatches in " + a.trace)
    if a.stamps:
        with open(a.stamps) as f:
            txt = f.read()
        st = json.loads(txt[txt.index("{"):])["stamps"][-1]
        t0 = t1 = None
        for clk in ("boottime_ns", "monotonic_ns", "realtime_ns"):
            c0, c1 = st["t0"][clk], st["t1"][clk]
            if any(c0 <= s_ <= c1 for s_, _, _ in ev):
                t0, t1 = c0, c1
                break
        if t0 is None:
            sys.exit("no clock of the stamps brackets any kernel of the trace")
        # the window is the proof's whole wall time on the host: GPU idle at either end (transcript, set-up of the call) counts
    elif a.window:
        t0, t1 = (int(x) for x in a.window.split(","))
    else:
        t0, t1 = cut_last_rep(ev, a.reps)
    r = analyse(ev, t0, t1, a.primary)
    md = to_md(r, a.title or a.trace)
    print(md)
    if a.gaps:
        win = sorted((s_, e_, n_) for s_, e_, n_ in ev if e_ > t0 and s_ < t1)
        gaps = []
        cover, last = t0, "(window start)"
        for s_, e_, n_ in win:
            if s_ > cover:
                gaps.append((s_ - cover, cover - t0, last, n_))
            if e_ > cover:
                cover, last = e_, n_
        if t1 > cover:
            gaps.append((t1 - cover, cover - t0, last, "(window end)"))
        gaps.sort(reverse=True)
        print(f"largest {a.gaps} idle gaps (ms, at ms into the window, kernel before -> kernel after); {len(gaps)} gaps in all")
        for g_, at, b_, n_ in gaps[: a.gaps]:
            print(f"  {g_ / 1e6:8.3f}  @{at / 1e6:9.3f}  {b_} -> {n_}")
    if a.md:
        with open(a.md, "a") as f:
            f.write(md + "\n")
    if a.json:
        with open(a.json, "w") as f:
            json.dump(r, f, indent=1)


if __name__ == "__main__":
    main()
