#!/usr/bin/env python3
"""Instruction-class counts per kernel of a hipcc -S --cuda-device-only listing (static counts; loops are not weighted).
Usage: python tools/isa_count.py file.s [name-filter]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
ends = [i for i, l in enumerate(lines) if "s_endpgm" in l]
for i, name in starts:
    if flt not in name:
        continue
    later = [x for x in ends if x > i]
    if not later:
        continue
    body = [l.strip() for l in lines[i + 1:later[0]]]
    body = [l for l in body if l and not l.startswith(";") and not l.startswith(".")]
    cnt = lambda pat: sum(1 for l in body if re.match(pat, l))
    valu = sum(1 for l in body if l.startswith("v_") and not l.startswith("v_mfma") and not l.startswith("v_accvgpr"))
    print(f"{name[:60]}: lines={len(body)} mfma={cnt('v_mfma')} valu={valu} accvgpr_mov={cnt('v_accvgpr')} scratch_ld={cnt('scratch_load')} "
          f"scratch_st={cnt('scratch_store')} ds_read={cnt('ds_read')} ds_write={cnt('ds_write')} global_ld={cnt('global_load')} "
          f"global_st={cnt('global_store')} salu={cnt('s_')} waitcnt={cnt('s_waitcnt')} barrier={cnt('s_barrier')}")
