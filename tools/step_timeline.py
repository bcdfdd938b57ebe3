#!/usr/bin/env python3
"""Timeline of ONE step out of a rocprofv3 --kernel-trace result (the rocpd sqlite database, or the *_kernel_trace.csv):
every kernel between two consecutive launches of a marker kernel, with its duration and the idle gap before it, and the
totals - how much of a step is kernels and how much is gaps between them.

    python tools/step_timeline.py gpurun_out/prof_nt adam_pack_kernel            (directory is searched for the db / csv)
"""
import csv
import glob
import os
import sqlite3
import sys


def load(path):
    dbs = glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
    if dbs:
        cur = sqlite3.connect(dbs[0]).cursor()
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
        kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
        ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
        return list(cur.execute(f"select d.start, d.end, s.kernel_name, d.grid_size_x, d.grid_size_y from {kd} d join {ks} s "
                                f"on d.kernel_id = s.id order by d.start"))
    rows = []
    for f in glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_X", 0) or 0),
                         int(r.get("Grid_Size_Y", 0) or 0)))
    return sorted(rows)


def main():
    rows = load(sys.argv[1])
    marker = sys.argv[2]
    idx = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(idx) < 3:
        sys.exit(f"marker {marker!r} seen {len(idx)} times in {len(rows)} dispatches")
    a, b = idx[-3], idx[-2]
    prev = rows[a][1]
    kernels = gaps = 0.0
    by = {}
    for r in rows[a + 1:b + 1]:
        gap, dur = (r[0] - prev) / 1e3, (r[1] - r[0]) / 1e3
        kernels += dur
        gaps += max(gap, 0.0)
        short = r[2].split("(")[0][-60:]
        by.setdefault(short, [0, 0.0])
        by[short][0] += 1
        by[short][1] += dur
        if "-v" in sys.argv:
            print(f"gap {gap:8.1f} us  dur {dur:9.1f} us  {short} grid {r[3]}x{r[4]}")
        prev = max(prev, r[1])
    step = (rows[b][1] - rows[a][1]) / 1e3
    for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print(f"{t:10.1f} us {100 * t / step:5.1f} %  x{n:<4d} {k}")
    print(f"step {step:.1f} us = kernels {kernels:.1f} us + gaps {gaps:.1f} us ({100 * gaps / step:.1f} %) over {b - a} launches")


if __name__ == "__main__":
    main()
