#!/usr/bin/env python3
"""Per-dispatch durations of ONE score evaluation of the NICE flow (51 launches) from a rocprofv3 kernel trace:
    cd /tmp && rocprofv3 --kernel-trace -d gpurun_out/nice_trace -- python tools/nice_gemm_trace.py run 4096 ; python tools/nice_gemm_trace.py report <db>"""
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    import torch

    from sde_sampler_amd import problems

    B = int(sys.argv[2])
    target = problems.build_target(dict(kind="nice", dim=196)).to("cuda:0")
    x = torch.randn(B, 196, device="cuda:0")
    for _ in range(4):
        target.score(x)
    torch.cuda.synchronize()
else:
    con = sqlite3.connect(sys.argv[2])
    rows = list(con.execute("select name, start, end, duration, grid_x, grid_y from kernels order by start"))
    rows = [r for r in rows if "nice" in r[0]]
    last = rows[-51:]  # the last evaluation
    t0 = last[0][1]
    prev_end = None
    for name, start, end, dur, gx, gy in last:
        gap = 0 if prev_end is None else start - prev_end
        short = "gemm<T>" if "Lb1" in name or "<true>" in name else ("gemm<N>" if "gemm" in name else name.split("(")[0].split("::")[-1])
        print(f"{short:22s} grid=({gx // 512 if 'gemm' in name else gx},{gy})  start {1e-3 * (start - t0):8.1f} us  dur {1e-3 * dur:6.1f} us  gap before {1e-3 * gap:5.1f} us")
        prev_end = end
    print(f"total {1e-3 * (last[-1][2] - t0):.1f} us; kernels {1e-3 * sum(r[3] for r in last):.1f} us")
