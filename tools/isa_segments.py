#!/usr/bin/env python3
"""Instruction classes of a kernel per barrier-delimited segment of its listing (hipcc -S --cuda-device-only): where the matrix
instructions, the AGPR <-> VGPR moves and the scratch accesses sit.   python tools/isa_segments.py file.s mangled-name-prefix"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith(sys.argv[2])][0]
end = [i for i, l in enumerate(lines) if "s_endpgm" in l and i > start][0]
body = [l.strip() for l in lines[start:end]]
body = [l for l in body if l and not l.startswith(";") and not l.startswith(".")]
seg, cur = [], []
for l in body:
    cur.append(l)
    if l.startswith("s_barrier"):
        seg.append(cur)
        cur = []
seg.append(cur)
for k, sg in enumerate(seg):
    c = lambda p: sum(1 for l in sg if re.match(p, l))
    valu = sum(1 for l in sg if l.startswith("v_") and not l.startswith("v_mfma") and not l.startswith("v_accvgpr"))
    print(f"{k:3d} len {len(sg):5d} mfma {c('v_mfma'):4d} accvgpr {c('v_accvgpr'):4d} valu {valu:5d} exp {c('v_exp'):3d} pk {c('v_pk'):4d} ds_read {c('ds_read'):4d} "
          f"ds_write {c('ds_write'):4d} scratch_ld {c('scratch_load'):4d} scratch_st {c('scratch_store'):4d} global {c('global_'):4d} waitcnt {c('s_waitcnt'):4d} "
          f"branches {c('s_cbranch|s_branch'):3d}")
