#!/usr/bin/env python3
"""Kernel timeline of ONE replayed optimisation step (tools/train_demo.py --graph under rocprofv3 --kernel-trace): every dispatch of the
last complete step with its start offset, duration and the idle gap in front of it.
Usage: python tools/graph_step_trace.py <results.db> [anchor kernel substring = traj_ws_kernel]"""
import sqlite3
import sys


def main(path, anchor="traj_ws_kernel"):
    con = sqlite3.connect(path)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if anchor in r[0]]
    if len(idx) < 3:
        print("not enough steps in the trace")
        return
    # a step = from one forward trajectory launch to the next; show the one before the last
    a, b = idx[-3], idx[-2]
    # the step really begins at the prior sample / prep in front of the forward kernel: walk back over the small kernels since the previous step's end
    t0 = rows[a][1]
    busy = 0
    print(f"# one optimisation step: {b - a} dispatches, {(rows[b][1] - t0) / 1e3:.1f} us from forward launch to forward launch")
    print(f"# {'start_us':>9} {'dur_us':>8} {'gap_us':>7}  kernel")
    prev_end = rows[a - 1][2] if a > 0 else t0
    for r in rows[a:b]:
        busy += r[2] - r[1]
        print(f"  {(r[1] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:8.2f} {(r[1] - prev_end) / 1e3:7.2f}  {r[0][:110]}")
        prev_end = r[2]
    print(f"# busy {busy / 1e3:.1f} us, idle {(rows[b][1] - t0 - busy) / 1e3:.1f} us")


if __name__ == "__main__":
    main(*sys.argv[1:])
