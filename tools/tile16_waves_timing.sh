#!/bin/bash
# teams of two against teams of four wavefronts (csrc/sdeh_bwdf16.hip) at the batch sizes around the launcher's thresholds
for spec in cfg2_gmm2_dis_kl cfg3_gmm50_pis_kl; do
  for waves in 2 4; do
    echo "== SDEH_BWD_TILE=16 SDEH_BWD_WAVES=$waves"; SDEH_BWD_TILE=16 SDEH_BWD_WAVES=$waves python tools/bwd_timing.py $spec kl 2048 4096 6144 8192 16384 2>&1 | grep -v amdgpu.ids | cut -c1-150
  done
  echo "== SDEH_BWD_TILE=32"; SDEH_BWD_TILE=32 python tools/bwd_timing.py $spec kl 6144 8192 16384 2>&1 | grep -v amdgpu.ids | cut -c1-150
done
