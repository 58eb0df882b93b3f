#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace) into a per-kernel stats table,
the same columns `--stats` prints (calls, total, average, min, max, percentage).
Usage: python tools/rocpd_summary.py <results.db> [> profiles/<name>.md]"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace summary of {path}\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, calls, tot, avg, mn, mx in rows:
        short = name.split("(")[0]
        print(f"| `{short}` | {calls} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100.0 * tot / total:.1f} |")
    try:
        rows = cur.execute("select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, grid_size, workgroup_size from kernels group by name").fetchall()
        print("\n| kernel | VGPR | AGPR | SGPR | LDS B | scratch B | grid | workgroup |")
        print("|---|---:|---:|---:|---:|---:|---:|---:|")
        for r in rows:
            print("| `" + r[0].split("(")[0] + "` | " + " | ".join(str(x) for x in r[1:]) + " |")
    except sqlite3.Error:
        pass


if __name__ == "__main__":
    main(sys.argv[1])
