#!/usr/bin/env python3
"""Kernel timeline out of a rocprofv3 rocpd database: every dispatch of at least --min-ms, ordered by start, with its queue and the
overlap with the previous line -- to see which kernels of a multi-stream schedule really ran side by side.
Usage: timeline_rocpd.py results.db [--min-ms 1.0] [--last 60]"""
import argparse
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db"); ap.add_argument("--min-ms", type=float, default=1.0); ap.add_argument("--last", type=int, default=60)
a = ap.parse_args()
c = sqlite3.connect(a.db)
rows = c.execute("select name, queue_id, stream_id, start, end, grid_x, workgroup_x from kernels where duration >= ? order by start", (a.min_ms * 1e6,)).fetchall()
rows = rows[-a.last:]
t0 = rows[0][3] if rows else 0
print("| start ms | end ms | ms | queue | stream | kernel | running at its start |")
print("|---|---|---|---|---|---|---|")
for i, r in enumerate(rows):
    n = r[0].replace("(anonymous namespace)::", "").replace("void ", "")
    n = n if len(n) < 60 else n[:57] + "..."
    live = [q[0].replace("(anonymous namespace)::", "").replace("void ", "")[:28] for q in rows[:i] if q[4] > r[3]]
    print("| %.2f | %.2f | %.2f | %s | %s | `%s` | %s |" % ((r[3] - t0) / 1e6, (r[4] - t0) / 1e6, (r[4] - r[3]) / 1e6, r[1], r[2], n, ", ".join(live) or "-"))
