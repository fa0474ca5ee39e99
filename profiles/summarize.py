#!/usr/bin/env python3
"""profiles/summarize.py <gpurun_out/prof_TAG> — print per-kernel averages from the rocprofv3
rocpd databases run_profile.sh leaves behind (kernel-trace stats + PMC passes)."""
import sqlite3
import sys
from pathlib import Path

d = Path(sys.argv[1])
for db in sorted(d.glob("*/*_results.db")):
    c = sqlite3.connect(str(db))
    print(f"== {db.parent.name}")
    if db.parent.name == "stats":
        for r in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
            print(f"  {r[0][:90]}  calls={r[1]} total_us={r[2]:.1f} avg_us={r[3]:.1f} pct={r[4]:.1f}")
    else:
        q = "select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"
        for r in c.execute(q):
            print(f"  {r[0][:60]:60s} {r[1]:28s} avg={r[2]:.1f} n={r[3]}")
