#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite output) results .db into the per-kernel stats table that
`rocprofv3 --kernel-trace --stats` describes: name, calls, total / average / min / max duration (ns), share.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.csv
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by 3 desc"
    ).fetchall() if _has_duration(c) else None
    if rows is None:
        rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                         "group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
    for n, calls, total, avg, mn, mx in rows:
        print(f"\"{n}\",{calls},{int(total)},{avg:.1f},{int(mn)},{int(mx)},{100.0 * total / tot:.3f}")


def _has_duration(c):
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    return "duration" in cols


if __name__ == "__main__":
    main(sys.argv[1])
