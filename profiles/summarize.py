#!/usr/bin/env python3
"""profiles/summarize.py <gpurun_out/prof_TAG> [--pmc-json OUT --workload W --frames N --kernel SUBSTR]
Print per-kernel averages from the rocprofv3 rocpd databases run_profile.sh leaves behind
(kernel-trace stats + PMC passes).  With --pmc-json, also write the HBM traffic per launch of the
kernel whose name contains SUBSTR (FETCH_SIZE / WRITE_SIZE from their own passes, KB = 1024 B,
FETCH_SIZE x2 on gfx950 as MI355X_MICROARCH.md prescribes) — the file bench.py reads `roofline.traffic` from."""
import argparse
import json
import sqlite3
from pathlib import Path

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--pmc-json")
ap.add_argument("--workload", default="cfg2")
ap.add_argument("--frames", type=int, default=256)
ap.add_argument("--kernel", default="tick_yuv_bgra_tiled")
ap.add_argument("--name", default="tick_nv12_bgra_tiled")
ap.add_argument("--source", default="")
a = ap.parse_args()

d = Path(a.dir)
pmc = {}
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
            if a.kernel in r[0]:
                pmc[r[1]] = r[2]

if a.pmc_json and "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    out = {
        "workload": a.workload, "frames": a.frames, "kernel": a.name,
        "fetch_size_kb": pmc["FETCH_SIZE"], "write_size_kb": pmc["WRITE_SIZE"],
        "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section); KB = 1024 B; separate --pmc passes",
        "hbm_bytes_per_launch": (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0,
        "source": a.source,
    }
    Path(a.pmc_json).write_text(json.dumps(out, indent=1))
    print("wrote", a.pmc_json)
