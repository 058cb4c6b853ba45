#!/usr/bin/env python3
"""Per-kernel average of every collected PMC counter from a rocprofv3 rocpd .db.
    python tools/rocpd_pmc.py x_results.db"""
import sqlite3, sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
print("# columns:", cols, file=sys.stderr)
name_col = "kernel_name" if "kernel_name" in cols else "name"
grid_col = "grid_size" if "grid_size" in cols else "0"
rows = c.execute(f"select {name_col}, counter_name, value, {grid_col} from counters_collection").fetchall()
acc = defaultdict(lambda: [0.0, 0])
for k, cn, v, g in rows:
    short = k.split("(")[0].replace("void wh::", "").replace("wh::", "")
    if "dec32_proj_kernel<2" in short:       # oproj / coproj / fc2 share one instantiation: the launch grid (fc2 splits K 4 ways) tells them apart
        short += f"@grid{g}"
    a = acc[(short, cn)]; a[0] += v; a[1] += 1
print("Kernel,Counter,Dispatches,AvgValue,Total")
for (k, cn), (tot, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"\"{k}\",{cn},{n},{tot / n:.1f},{tot:.0f}")
