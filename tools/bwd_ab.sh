#!/bin/bash
# A/B of the fused training backward: trajectory-split teams (sdeh_bwdf2.hip) against channel-split teams (SDEH_BWD_V1=1), kernel times
# at B = 65 536 (and 16 384 / 32 768) for the BASELINE shapes.   tools/bwd_ab.sh > gpurun_out/bwd_ab.txt
cd "$(dirname "$0")/.."
for v in "" 1; do
  if [ -n "$v" ]; then export SDEH_BWD_V1=1; echo "== channel-split (SDEH_BWD_V1=1)"; else unset SDEH_BWD_V1; echo "== trajectory-split"; fi
  REPS=${REPS:-7} python tools/bwd_timing.py cfg2_gmm2_dis_kl kl 16384 65536
  REPS=${REPS:-7} python tools/bwd_timing.py cfg3_gmm50_pis_kl kl 16384 65536
  REPS=${REPS:-7} python tools/bwd_timing.py cfg1_dw_dis_lv lv 2048 65536
  REPS=${REPS:-7} python tools/bwd_timing.py cfg1_dw_dis_lv kl 65536
  REPS=${REPS:-5} python tools/bwd_timing.py cfg4_funnel_dds_lv lv 65536
done
