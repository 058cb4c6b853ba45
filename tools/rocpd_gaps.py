#!/usr/bin/env python3
"""Per-kernel duration AND the idle gap in front of each kernel (start - previous end on the same device) from a
rocprofv3 rocpd .db: separates kernel time from launch/boundary time in a chain of short dependent kernels.

    python tools/rocpd_gaps.py x_results.db [name-substring]
"""
import sqlite3
import sys
from collections import defaultdict


def main(path, sub=None):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    dur, gap = defaultdict(list), defaultdict(list)
    prev_end = None
    for name, s, e in rows:
        short = name.split("(")[0].replace("void wh::", "").replace("wh::", "")
        dur[short].append(e - s)
        if prev_end is not None:
            gap[short].append(s - prev_end)
        prev_end = e
    print(f"{'kernel':58s} {'calls':>7s} {'avg_ns':>9s} {'p50_ns':>9s} {'min_ns':>8s} {'gap_before_p50':>14s} {'gap_avg':>9s}")
    tot = 0
    for k in sorted(dur, key=lambda k: -sum(dur[k])):
        if sub and sub not in k:
            continue
        d = sorted(dur[k]); g = sorted(gap[k]) or [0]
        tot += sum(d)
        print(f"{k[:58]:58s} {len(d):7d} {sum(d) / len(d):9.0f} {d[len(d) // 2]:9d} {d[0]:8d} {g[len(g) // 2]:14d} {sum(g) / len(g):9.0f}")
    span = rows[-1][2] - rows[0][1]
    print(f"sum of kernel time {tot / 1e6:.3f} ms; first start -> last end {span / 1e6:.3f} ms")


if __name__ == "__main__":
    main(*sys.argv[1:])
