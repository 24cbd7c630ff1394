#!/usr/bin/env python3
"""Turns a rocprofv3 (ROCm 7.x, rocpd sqlite output) results database into the plain-text per-kernel summary
committed under profiles/.  Usage: python tools/rocprof_summary.py <results.db> [> profiles/rNN_xxx.txt]"""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    print(f"# rocprofv3 --kernel-trace --stats summary of {path.split('/')[-1]}")
    print(f"# {'calls':>6} {'total_us':>12} {'avg_us':>12} {'pct':>7}  kernel")
    for name, calls, total, avg, pct in con.execute(
            "select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc limit 25"):
        print(f"  {calls:6d} {total:12.1f} {avg:12.2f} {pct:7.2f}  {name[:150]}")
    print("# dispatch details of the dominant kernel")
    row = con.execute("select name from top_kernels order by total_duration desc limit 1").fetchone()
    cur = con.execute("select duration, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count "
                      "from kernels where name = ? order by start", (row[0],))
    rows = cur.fetchall()
    for r in rows[:40]:
        print("  duration_ns=%d grid=%d wg=%d lds=%d scratch=%d vgpr=%d agpr=%d sgpr=%d" % r)
    if len(rows) > 40:  # (a stepped evaluation launches its segment kernel thousands of times)
        ds = sorted(r[0] for r in rows)
        print(f"  ... {len(rows) - 40} more dispatches; all {len(rows)}: min {ds[0]} ns, median {ds[len(ds) // 2]} ns, max {ds[-1]} ns")


if __name__ == "__main__":
    main(sys.argv[1])
