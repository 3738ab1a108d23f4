#!/usr/bin/env python3
"""Dump the per-kernel summary (the `--stats` view) of a rocprofv3 rocpd .db as text.
usage: python tools/rocpd_summary.py <results.db> [> profiles/<name>.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
print(f"# rocprofv3 --kernel-trace --stats summary of {sys.argv[1]}")
print(f"{'calls':>6} {'total_us':>12} {'avg_us':>12} {'pct':>7}  kernel")
for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f"{calls:>6} {total:>12.1f} {avg:>12.2f} {pct:>7.2f}  {name[:110]}")
try:
    rows = db.execute("select name, count(*), avg(vgpr_count), avg(accum_vgpr_count), avg(sgpr_count), avg(lds_size), avg(scratch_size), avg(grid_x), avg(workgroup_x) "
                      "from kernels group by name order by sum(duration) desc limit 6").fetchall()
    print("\n# resources (vgpr, agpr, sgpr, lds bytes, scratch bytes/lane, grid_x, wg_x)")
    for r in rows:
        print(f"{r[0][:70]:70s} vgpr={r[2]:.0f} agpr={r[3]:.0f} sgpr={r[4]:.0f} lds={r[5]:.0f} scratch={r[6]:.0f} grid={r[7]:.0f} wg={r[8]:.0f}")
except Exception as e:  # pragma: no cover
    print("# (no per-dispatch resource columns)", e)
