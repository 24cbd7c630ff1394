#!/bin/bash
# Per-phase cycle counts of the fused backward kernel (one wave: block 0 / wave 0): builds a measurement copy of the library with
# -DSDEH_BWDF_PROFILE (s_memtime at the phase boundaries, never part of the shipped build) and runs tools/bwd_timing.py with it.
#   bash tools/bwdf_phase_profile.sh            (build here, where hipcc is; the .so travels to the GPU box under prof_tmp/)
#   bash tools/bwdf_phase_profile.sh run        (on the GPU box)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "${1:-build}" = "build" ]; then
  mkdir -p $ROOT/prof_tmp
  cd $ROOT/sde_sampler_amd/csrc
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=gfx950 -Wno-comment -DSDEH_BWDF_PROFILE -c sdeh_bwdf.hip -o /tmp/bwdf_prof.o &
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=gfx950 -Wno-comment -DSDEH_BWDF_PROFILE -c sdeh_bwdf2.hip -o /tmp/bwdf2_prof.o
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v "sdeh_bwdf.o\|sdeh_bwdf2.o") /tmp/bwdf_prof.o /tmp/bwdf2_prof.o -o $ROOT/prof_tmp/libsdeh_prof.so
  echo "built $ROOT/prof_tmp/libsdeh_prof.so"
else
  cd $ROOT
  for c in "cfg2_gmm2_dis_kl kl 65536" "cfg2_gmm2_dis_kl kl 2048" "cfg3_gmm50_pis_kl kl 65536" "cfg1_dw_dis_lv lv 65536"; do
    REPS=1 SDEH_LIBRARY=$ROOT/prof_tmp/libsdeh_prof.so python tools/bwd_timing.py $c 2>&1 | grep -E "phases|backward" | tail -2
  done
fi
