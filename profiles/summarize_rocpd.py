#!/usr/bin/env python3
"""Turns a rocprofv3 rocpd database (rocprofv3 --kernel-trace --stats ... -> *_results.db) into the per-kernel
summary table committed under profiles/.  Usage: summarize_rocpd.py results.db > summary.md"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                 "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
                 "from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print("| kernel | calls | total ms | avg ms | min ms | max ms | % | VGPR | SGPR | LDS B | grid.x | wg.x |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for r in rows:
    n = r[0] if len(r[0]) < 90 else r[0][:87] + "..."
    print("| `%s` | %d | %.3f | %.3f | %.3f | %.3f | %.1f | %s | %s | %s | %s | %s |" %
          (n, r[1], r[2] / 1e6, r[3] / 1e6, r[4] / 1e6, r[5] / 1e6, 100.0 * r[2] / tot, r[6], r[7], r[8], r[9], r[10]))
