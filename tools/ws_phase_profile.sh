#!/bin/bash
# Cycle counts of the M and the V wave of one trajectory group of the wave-specialised kernel (headline variant): builds a
# measurement copy of the library with -DSDEH_WS_PROFILE (s_memtime at the hand-off points; never part of the shipped build).
#   bash tools/ws_phase_profile.sh          (build here; the .so travels to the GPU box under prof_tmp/)
#   bash tools/ws_phase_profile.sh run [B]  (on the GPU box; B = 32768: the groups of 32 trajectories)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "${1:-build}" = "build" ]; then
  mkdir -p $ROOT/prof_tmp
  cd $ROOT/sde_sampler_amd/csrc
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=gfx950 -Wno-comment -DSDEH_WS_PROFILE -DSDEH_DP=50 -DSDEH_PAD=0 \
    -DSDEH_SPECNAME=pis_gmm4 -DSDEH_SPEC="1,1,1,2,0,0" -DSDEH_GENERIC=0 -DSDEH_GNV=4 -c sdeh_traj_inst.hip -o /tmp/traj_prof.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v traj_50_0_pis_gmm4.o) /tmp/traj_prof.o -o $ROOT/prof_tmp/libsdeh_wsprof.so
  echo "built $ROOT/prof_tmp/libsdeh_wsprof.so"
else
  cd $ROOT
  SDEH_LIBRARY=$ROOT/prof_tmp/libsdeh_wsprof.so python tools/quick_time.py 12 ${2:-65536} 2>&1 | grep -E "phases|kernel ms" | tail -3
fi
