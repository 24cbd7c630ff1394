#!/bin/bash
# rocprofv3 PMC passes over the headline workload (run on the GPU box, from the repo root).
# Counters are collected in separate passes WITHOUT trace domains (only --kernel-trace), as the pool requires.
set -u
OUT=${1:-gpurun_out/pmc}
# optional: the command to profile (default: 4 launches of the headline workload), e.g.
#   tools/pmc_profile.sh gpurun_out/pmc_wide "python tools/wide_timing.py wide_pis_funnel196 32768"
CMD=${2:-}
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  if [ -z "$CMD" ]; then
    timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d "$ROOT/$OUT/$name" -- python "$ROOT/tools/quick_time.py" 4 > "$ROOT/$OUT/$name.log" 2>&1
  else
    (cd "$ROOT" && timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d "$ROOT/$OUT/$name" -- $CMD > "$ROOT/$OUT/$name.log" 2>&1)
  fi
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
run sq3 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run fetch FETCH_SIZE
run write WRITE_SIZE
cd "$ROOT"
python tools/pmc_summary.py "$OUT" ${KERNEL:-} | tee "$OUT/summary.txt"   # KERNEL=bwdf_kernel: another kernel's counters
# the raw rocprofv3 databases are large (gpurun_out is capped at 64 MiB): keep only the summary and logs
find "$OUT" -name "*.db" -delete
