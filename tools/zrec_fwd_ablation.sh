#!/bin/bash
# What the pre-activation record costs the training forward (headline variant, d = 50): measurement copies of the library with the M / V
# wave cycle counters (-DSDEH_WS_PROFILE) and (a) the shipped non-temporal stores, (b) plain stores, (c) no record stores at all.
#   bash tools/zrec_fwd_ablation.sh            (build here; the .so files travel to the GPU box under prof_tmp/)
#   bash tools/zrec_fwd_ablation.sh run        (on the GPU box)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "${1:-build}" = "build" ]; then
  mkdir -p $ROOT/prof_tmp
  cd $ROOT/sde_sampler_amd/csrc
  for v in nt:"" nont:-DSDEH_ZREC_NO_NT skip:-DSDEH_ZREC_SKIP; do
    tag=${v%%:*}; flag=${v#*:}
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=gfx950 -Wno-comment -DSDEH_WS_PROFILE $flag ${EXTRA:-} -DSDEH_DP=50 -DSDEH_PAD=0 \
      -DSDEH_SPECNAME=pis_gmm4 -DSDEH_SPEC="1,1,1,2,0,0" -DSDEH_GENERIC=0 -DSDEH_GNV=4 -c sdeh_traj_inst.hip -o /tmp/traj_zabl_$tag.o &
  done
  wait
  for tag in nt nont skip; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v traj_50_0_pis_gmm4.o) /tmp/traj_zabl_$tag.o -o $ROOT/prof_tmp/libsdeh_zabl_$tag.so
  done
  echo "built $ROOT/prof_tmp/libsdeh_zabl_{nt,nont,skip}.so"
else
  cd $ROOT
  for tag in nt nont skip; do
    echo "== record stores: $tag"
    REPS=3 SDEH_LIBRARY=$ROOT/prof_tmp/libsdeh_zabl_$tag.so python tools/zrec_ab.py ${2:-cfg3_gmm50_pis_kl:lv:65536} 2>&1 | grep -E "phases|record fwd" | tail -4
  done
fi
