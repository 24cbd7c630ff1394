#!/bin/bash
# PMC passes over the fused training backward of BASELINE configs[1] (GMM-40 d = 2, DIS kl, B = 65 536, T = 100) with and without the
# pre-activation record: executed matrix FLOPs (SQ_INSTS_VALU_MFMA_MOPS_F32 / SQ_INSTS_MFMA) against the algorithmic 2 x (4 d C + 2 Lh C^2)
# per trajectory-step, matrix / vector pipe busy, HBM bytes.  Separate --pmc passes with --kernel-trace only (the pool's rule).
#   bash tools/pmc_bwd.sh          (on the GPU box; writes gpurun_out/pmc_bwd_{zrec,reeval}/summary.txt)
set -u
ROOT=$(pwd)
for mode in zrec reeval; do
  OUT=gpurun_out/pmc_bwd_$mode
  mkdir -p $OUT
  if [ $mode = reeval ]; then export SDEH_BWD_ZREC=0; else unset SDEH_BWD_ZREC; fi
  CMD="python $ROOT/tools/bwd_timing.py cfg2_gmm2_dis_kl kl 65536"
  cd /tmp && export TMPDIR=/tmp
  for pass in "sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
              "sq3 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU" \
              "grbm GRBM_GUI_ACTIVE GRBM_COUNT" "fetch FETCH_SIZE" "write WRITE_SIZE"; do
    set -- $pass
    name=$1; shift
    REPS=1 timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d "$ROOT/$OUT/$name" -- $CMD > "$ROOT/$OUT/$name.log" 2>&1
  done
  cd $ROOT
  python tools/pmc_summary.py $OUT bwdf2_kernel | tee $OUT/summary.txt
  find $OUT -name "*.db" -delete
done
