#!/usr/bin/env python3
"""Summarises the rocprofv3 --pmc passes written by tools/pmc_profile.sh: per-dispatch counter values of the
trajectory kernel (summed over shader engines / XCDs as rocprofv3 reports them), averaged over dispatches."""
import glob
import sqlite3
import sys


def main(out, pattern=None):
    like = (f"{{c}} like '%{pattern}%'" if pattern else "({c} like '%traj%' or {c} like '%bridge_wide_kernel%')")
    for db in sorted(glob.glob(f"{out}/*/**/*.db", recursive=True)):
        con = sqlite3.connect(db)
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
        view = "counters_collection" if "counters_collection" in tabs else None
        if view is None:
            print(db, "no counters_collection view; tables:", [t for t in tabs if "pmc" in t or "counter" in t])
            continue
        cols = [d[1] for d in con.execute(f"pragma table_info({view})")]
        namecol = "name" if "name" in cols else ("kernel_name" if "kernel_name" in cols else None)
        rows = con.execute(f"select counter_name, avg(value), count(*) from (select dispatch_id, counter_name, sum(value) as value "
                           f"from {view} where {like.format(c=namecol)} group by dispatch_id, counter_name) group by counter_name").fetchall()
        print(f"# {db.split('/')[-3]}")
        for name, val, n in rows:
            print(f"  {name:34s} {val:20.1f}   (avg over {n} dispatches)")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)  # optional: kernel-name pattern (default: the trajectory kernels)
