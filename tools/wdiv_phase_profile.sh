#!/bin/bash
# Per-phase cycle counts of the wide Bridge divergence backward (one wave: block 0 / wave 0): builds a measurement copy of the library
# with -DSDEH_WDIV_PROFILE (never part of the shipped build) and runs tools/wide_train_timing.py with it.
#   bash tools/wdiv_phase_profile.sh            (build here, where hipcc is; the .so travels to the GPU box under prof_tmp/)
#   bash tools/wdiv_phase_profile.sh run        (on the GPU box)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "${1:-build}" = "build" ]; then
  mkdir -p $ROOT/prof_tmp
  cd $ROOT/sde_sampler_amd/csrc
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=gfx950 -Wno-comment -DSDEH_WDIV_PROFILE -c sdeh_wide_bwd.hip -o /tmp/wdiv_prof.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v sdeh_wide_bwd.o) /tmp/wdiv_prof.o -o $ROOT/prof_tmp/libsdeh_wdiv_prof.so
  echo "built $ROOT/prof_tmp/libsdeh_wdiv_prof.so"
else
  cd $ROOT
  SDEH_LIBRARY=$ROOT/prof_tmp/libsdeh_wdiv_prof.so python tools/wide_train_timing.py cfg5_like_bridge196 ${2:-4096} lv ${3:-10} 2>&1 | grep -E "wdiv|rep 2"
fi
