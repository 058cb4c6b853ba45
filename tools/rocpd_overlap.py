#!/usr/bin/env python3
"""Concurrency picture of a multi-stream run from a rocprofv3 rocpd .db (kernel trace): how long 0 / 1 / 2 / 3+ kernels were running,
how long at least one HBM-bound cross-attention kernel / one encoder GEMM was running, and per queue the busy time and the idle gaps
between consecutive kernels - what the in-flight regime of bench.py loses against the sum of its kernels.

    python tools/rocpd_overlap.py x_results.db [t0_fraction t1_fraction]     (analyse only the [t0, t1] fraction of the trace)
"""
import sqlite3
import sys
from collections import defaultdict


def main(path, f0=0.0, f1=1.0):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    qcol = next((k for k in ("stream_id", "queue_id", "queue", "stream") if k in cols), None)
    rows = c.execute(f"select name, start, end, {qcol or '0'} from kernels order by start").fetchall()
    t_lo, t_hi = rows[0][1], max(r[2] for r in rows)
    a, b = t_lo + (t_hi - t_lo) * float(f0), t_lo + (t_hi - t_lo) * float(f1)
    rows = [r for r in rows if r[1] >= a and r[2] <= b]
    ev = []
    for name, s, e, q in rows:
        kind = "xattn" if ("cross_attn" in name or "xabs_attn" in name) else "gemm" if ("gemm256" in name or "gemm_kernel" in name or "encoder_attention" in name) else "other"
        ev.append((s, 1, kind)); ev.append((e, -1, kind))
    ev.sort()
    run = defaultdict(int)
    hist, t_x, t_g, t_xg = defaultdict(int), 0, 0, 0
    prev = ev[0][0]
    n = 0
    for t, d, kind in ev:
        dt = t - prev
        if dt > 0:
            hist[min(n, 4)] += dt
            if run["xattn"] > 0: t_x += dt
            if run["gemm"] > 0: t_g += dt
            if run["xattn"] > 0 and run["gemm"] > 0: t_xg += dt
        n += d; run[kind] += d; prev = t
    span = ev[-1][0] - ev[0][0]
    # how many cross-attention kernels run at once (each takes slots x key splits CUs: 128 of 256 at 64 slots x 2 splits)
    xh, nx, prev = defaultdict(int), 0, ev[0][0]
    for t, d, kind in ev:
        if t > prev:
            xh[nx] += t - prev
        prev = t
        if kind == "xattn":
            nx += d
    print("  cross-attention kernels running at once: " + ", ".join(f"{k}: {100.0 * v / span:.1f} %" for k, v in sorted(xh.items())))
    dur = defaultdict(list)
    for name, s_, e_, q in rows:
        dur[name.split("(")[0][-60:]].append(e_ - s_)
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:12]:
        v.sort()
        print(f"  {k:60s} n {len(v):6d}  avg {sum(v) / len(v) / 1e3:8.1f} us  p50 {v[len(v) // 2] / 1e3:8.1f}  p90 {v[int(len(v) * 0.9)] / 1e3:8.1f}  total {sum(v) / 1e6:8.1f} ms")
    print(f"queue column: {qcol}; kernels {len(rows)}; span {span / 1e6:.2f} ms")
    for k in sorted(hist):
        print(f"  {k}{'+' if k == 4 else ' '} kernels running: {hist[k] / 1e6:9.2f} ms  {100.0 * hist[k] / span:5.1f} %")
    print(f"  >= 1 cross-attention running: {t_x / 1e6:.2f} ms ({100.0 * t_x / span:.1f} %);  >= 1 encoder GEMM / attention: {t_g / 1e6:.2f} ms "
          f"({100.0 * t_g / span:.1f} %);  both: {t_xg / 1e6:.2f} ms")
    # idle time in front of a kernel ON ITS OWN QUEUE (start - end of the queue's previous kernel), by the kind of the kernel that waited:
    # a dependent launch chain shows its kernel boundaries here, a launch that had to wait for CUs shows the wait
    last_end, wait = {}, defaultdict(lambda: [0, 0])
    for name, s_, e_, q in rows:
        if q in last_end:
            k = name.split("(")[0].replace("void wh::", "")[-44:]
            wait[k][0] += max(0, s_ - last_end[q]); wait[k][1] += 1
        last_end[q] = e_
    tot_wait = sum(v[0] for v in wait.values())
    print(f"  idle time in front of a kernel on its own queue, by kernel (total {tot_wait / 1e6:.1f} ms over {len(last_end)} queues):")
    for k, (w, n_) in sorted(wait.items(), key=lambda kv: -kv[1][0])[:10]:
        print(f"    {k:46s} n {n_:6d}  avg {w / max(n_, 1) / 1e3:7.1f} us  total {w / 1e6:8.1f} ms")
    per_q = defaultdict(list)
    for name, s, e, q in rows:
        per_q[q].append((s, e))
    for q, ks in sorted(per_q.items(), key=lambda kv: -len(kv[1]))[:8]:
        busy = sum(e - s for s, e in ks)
        gaps = sorted(max(0, ks[i + 1][0] - ks[i][1]) for i in range(len(ks) - 1))
        if not gaps:
            continue
        big = sum(g for g in gaps if g > 20000)
        print(f"  queue {q}: {len(ks)} kernels, busy {busy / 1e6:.2f} ms, gaps total {sum(gaps) / 1e6:.2f} ms (p50 {gaps[len(gaps) // 2]} ns, "
              f"p90 {gaps[int(len(gaps) * 0.9)]} ns, gaps > 20 us: {big / 1e6:.2f} ms)")


if __name__ == "__main__":
    main(*sys.argv[1:])
