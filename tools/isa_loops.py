"""Loop structure of one kernel's disassembly (llvm-objdump -d of a gfx950 code object): for every backward branch the instruction
count of the loop body and its MFMA / scratch / LDS / global-memory instructions -- where do the spills of a kernel live?
    python tools/isa_loops.py <disassembly.s> <mangled-kernel-name-substring>"""
import re
import sys

text = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(text) if re.match(r"^[0-9a-f]+ <", l) and key in l)
end = next((i for i in range(start + 1, len(text)) if re.match(r"^[0-9a-f]+ <", text[i])), len(text))
base = int(text[start].split()[0], 16)
ins = []  # (address, text)
for l in text[start + 1:end]:
    m = re.search(r"//\s*([0-9A-Fa-f]{12}):", l)
    if m:
        ins.append((int(m.group(1), 16), l.strip().split("//")[0].strip()))
index = {a: i for i, (a, _) in enumerate(ins)}
print(f"{len(ins)} instructions; scratch_load {sum('scratch_load' in t for _, t in ins)}, scratch_store {sum('scratch_store' in t for _, t in ins)}, "
      f"mfma {sum('v_mfma' in t for _, t in ins)}")
loops = []
for i, (a, t) in enumerate(ins):
    if t.startswith("s_cbranch") or t.startswith("s_branch"):
        m = re.search(r"\+0x([0-9a-f]+)>", text[start + 1 + i]) if False else None
for i, l in enumerate(text[start + 1:end]):
    if "s_cbranch" in l or "s_branch" in l:
        m = re.search(r"\+0x([0-9a-f]+)>", l)
        a = re.search(r"//\s*([0-9A-Fa-f]{12}):", l)
        if m and a:
            tgt, here = base + int(m.group(1), 16), int(a.group(1), 16)
            if tgt < here and tgt in index:
                loops.append((index[tgt], index[here]))
for s, e in sorted(loops):
    body = [t for _, t in ins[s:e + 1]]
    c = lambda k: sum(k in b for b in body)
    print(f"loop [{s:6d}, {e:6d}] {e - s + 1:6d} instr: mfma {c('v_mfma'):4d} scratch_load {c('scratch_load'):4d} scratch_store {c('scratch_store'):4d} "
          f"ds {c('ds_'):4d} global {c('global_'):4d} barrier {c('s_barrier'):3d}")
