"""Register / scratch / LDS table of every gfx950 kernel in the built objects (llvm-readelf --notes on the code objects that
llvm-objdump --offloading extracts):   python tools/kernel_resources.py [object ...]      (default: sde_sampler_amd/csrc/build/*.o)"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(.*", "", o).replace("sdeh::", "") for o in out]


def kernels(obj):
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(obj))
        os.symlink(os.path.abspath(obj), local)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], capture_output=True, cwd=tmp)
        cos = [p for p in glob.glob(local + ".*") if "amdgcn" in p]
        rows = []
        for co in cos:
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            for block in notes.split("- .agpr_count:")[1:]:
                f = lambda key: re.search(rf"\.{key}:\s+(\S+)", block)
                name = f("name")
                if not name:
                    continue
                agpr = int(block.split()[0])
                rows.append((name.group(1), int(f("vgpr_count").group(1)), agpr, int(f("private_segment_fixed_size").group(1)),
                             int(f("group_segment_fixed_size").group(1))))
        return rows


if __name__ == "__main__":
    objs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "sde_sampler_amd", "csrc", "build", "*.o")))
    print("# object | kernel | VGPRs (arch + acc) | AGPRs | scratch bytes per lane | static LDS bytes")
    for obj in objs:
        rows = kernels(obj)
        names = demangle([r[0] for r in rows])
        for (raw, vgpr, agpr, scratch, lds), name in sorted(zip(rows, names), key=lambda t: t[1]):
            print(f"{os.path.basename(obj):24s} {name[:110]:110s} {vgpr:4d} {agpr:4d} {scratch:6d} {lds:7d}")
