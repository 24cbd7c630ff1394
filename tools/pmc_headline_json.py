#!/usr/bin/env python3
"""profiles/pmc_headline.json from the summary of tools/pmc_profile.sh (headline workload), stamped with the hash of the headline
kernel's sources (bench.py: headline_kernel_sha) -- bench.py reports `roofline.traffic` / `executed_tflops` only while that hash
matches the tree it runs from.

    python tools/pmc_headline_json.py gpurun_out/r03/pmc_headline/summary.txt profiles/r03_pmc_headline.txt
"""
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

WANT = ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32",
        "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE")


def main(summary: str, source: str):
    rec = {}
    for line in Path(summary).read_text().splitlines():
        m = re.match(r"\s+(\S+)\s+([0-9.eE+-]+)\s+\(avg over (\d+) dispatches\)", line)
        if m and m.group(1) in WANT:
            key = m.group(1) + ("_KiB" if m.group(1) in ("FETCH_SIZE", "WRITE_SIZE") else "")
            rec[key] = float(m.group(2))
    missing = [k for k in ("FETCH_SIZE_KiB", "WRITE_SIZE_KiB", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_FMA_F32") if k not in rec]
    if missing:
        raise SystemExit(f"{summary}: counters missing: {missing}")
    rec["source"] = f"{source} (tools/pmc_profile.sh, separate --pmc passes, avg over the dispatches of the trajectory kernel)"
    rec["workload"] = "gmm50_pis_headline B=65536 T=100"
    rec["kernel_sha"] = bench.headline_kernel_sha()
    rec["kernel_sources"] = list(bench.HEADLINE_KERNEL_SOURCES)
    (ROOT / "profiles" / "pmc_headline.json").write_text(json.dumps(rec, indent=1) + "\n")
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
