for spec in cfg2_gmm2_dis_kl cfg3_gmm50_pis_kl; do
  for tile in 32 16; do
    echo "== SDEH_BWD_TILE=$tile"; SDEH_BWD_TILE=$tile python tools/bwd_timing.py $spec kl 512 2048 4096 8192 16384 2>&1 | grep -v amdgpu.ids
  done
done
