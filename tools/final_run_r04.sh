#!/bin/bash
# Round-4 measurement run (on the GPU box, from the repo root): everything whose summary is committed under profiles/r04_*.
#   gpurun --timeout 2400 -- 'bash tools/final_run_r04.sh'
set -u
OUT=gpurun_out/r04
mkdir -p $OUT
ROOT=$(pwd)
# 1. PMC passes of the headline trajectory kernel (separate --pmc passes, kernel trace only) + the stamped record bench.py reads
bash tools/pmc_profile.sh $OUT/pmc_headline > $OUT/pmc.log 2>&1
{ echo "# rocprofv3 PMC passes (tools/pmc_profile.sh, separate --pmc passes with --kernel-trace only) of the headline trajectory kernel traj_ws<50_0_pis_gmm4>,"
  echo "# GMM-40 d=50, B=65536, T=100, round 4; per launch, averaged over the dispatches.  GRBM_GUI_ACTIVE is summed over the 8 XCDs."
  cat $OUT/pmc_headline/summary.txt; } > $OUT/r04_pmc_headline.txt
python tools/pmc_headline_json.py $OUT/pmc_headline/summary.txt profiles/r04_pmc_headline.txt > /dev/null; cp profiles/pmc_headline.json $OUT/
# 2. rocprofv3 kernel trace of the bench command
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_headline -- python $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra > $ROOT/$OUT/prof_headline.log 2>&1)
DB=$(find $OUT/prof_headline -name "*.db" | head -1)
python tools/rocprof_summary.py $DB > $OUT/r04_kernel_stats_headline.txt
tail -2 $OUT/prof_headline.log | cut -c1-900 >> $OUT/r04_kernel_stats_headline.txt
find $OUT/prof_headline -name "*.db" -delete
# 3. the bench lines
python bench.py > $OUT/r04_bench_headline.json 2> $OUT/bench_headline.err
python bench.py --eager --no-extra --no-cpu-baseline > $OUT/r04_bench_headline_eager.json 2>> $OUT/bench_headline.err
for w in train_gmm2_dis_kl train_gmm50_pis_kl; do python bench.py --workload $w --no-cpu-baseline > $OUT/r04_bench_$w.json 2>> $OUT/bench_train.err; done
# 4. every BASELINE configuration at its per-GPU batch; the training backward at B = 65 536 / 2048 (both team splits)
python tools/all_configs_timing.py 2>&1 | grep -v amdgpu.ids > $OUT/r04_all_configs_timing.txt
{ REPS=9 python tools/bwd_timing.py cfg2_gmm2_dis_kl kl 2048 65536; REPS=9 python tools/bwd_timing.py cfg1_dw_dis_lv lv 2048 65536
  REPS=9 python tools/bwd_timing.py cfg1_dw_dis_lv kl 65536; REPS=7 python tools/bwd_timing.py cfg3_gmm50_pis_kl kl 2048 65536
  REPS=7 python tools/bwd_timing.py cfg3_gmm50_pis_kl lv 65536; REPS=7 python tools/bwd_timing.py cfg4_funnel_dds_lv lv 2048 65536
  echo "== through time at the reference's training batches: the scan form (d <= 4, default up to 3072 trajectories) against the 16-trajectory kernel (SDEH_BWD_SCAN=0)"
  REPS=9 python tools/bwd_timing.py cfg2_gmm2_dis_kl kl 512 2048; REPS=9 python tools/bwd_timing.py cfg1_dw_dis_lv kl 512 2048
  SDEH_BWD_SCAN=0 REPS=9 python tools/bwd_timing.py cfg2_gmm2_dis_kl kl 512 2048; SDEH_BWD_SCAN=0 REPS=9 python tools/bwd_timing.py cfg1_dw_dis_lv kl 512 2048
  echo "== channel-split teams (plan option SDEH_BWD_V1)"
  SDEH_BWD_V1=1 REPS=7 python tools/bwd_timing.py cfg2_gmm2_dis_kl kl 65536; SDEH_BWD_V1=1 REPS=7 python tools/bwd_timing.py cfg1_dw_dis_lv lv 65536
  SDEH_BWD_V1=1 REPS=5 python tools/bwd_timing.py cfg3_gmm50_pis_kl lv 65536; SDEH_BWD_V1=1 REPS=5 python tools/bwd_timing.py cfg4_funnel_dds_lv lv 65536
  echo "== trajectory-split teams forced for two coordinate tiles through time (plan option SDEH_BWD_V2)"
  SDEH_BWD_V2=1 REPS=5 python tools/bwd_timing.py cfg3_gmm50_pis_kl kl 65536; } 2>&1 | grep -v amdgpu.ids > $OUT/r04_training_backward_roofline.txt
# 5. phase profile of the trajectory-split backward (measurement build, prof_tmp/)
[ -f prof_tmp/libsdeh_prof.so ] && bash tools/bwdf_phase_profile.sh run 2>&1 | grep -v amdgpu.ids > $OUT/r04_bwd_fused_phases.txt
# 6. PMC + kernel trace of the trajectory-split backward (configs[1], B = 65 536)
KERNEL=bwdf2_kernel bash tools/pmc_profile.sh $OUT/pmc_bwdf2 "python tools/bwd_timing.py cfg2_gmm2_dis_kl kl 65536" > $OUT/pmc_bwdf2.log 2>&1
cp $OUT/pmc_bwdf2/summary.txt $OUT/r04_pmc_bwd_fused2.txt
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_train -- python $ROOT/bench.py --workload train_gmm2_dis_kl --steps 10 --warmup 3 --no-cpu-baseline > $ROOT/$OUT/prof_train.log 2>&1)
DB=$(find $OUT/prof_train -name "*.db" | head -1)
python tools/rocprof_summary.py $DB > $OUT/r04_kernel_stats_train_gmm2_dis_kl.txt
find $OUT/prof_train -name "*.db" -delete
# 6b. Bridge training step (conf/solver/bridge.yaml / basic_bridge.yaml with the shipped 64-channel networks): split forward + fused backwards
#     against the step-sequential forward (SDEH_BRIDGE_SEQ) and the plane kernels (SDEH_BWD_PLANES); per-kernel traces of two shapes
{ echo "# Bridge training step (loss + backward), T = 200, wall clock, eager, best of 5 (tools/bridge_step_profile.py <d> <B> <method>); MI355X, round 4"
  echo "# default: plain launch + row-parallel inference pass forward, sdeh_ctrl_backward_fused[_ex] + sdeh_bridge_backward_fused backward"
  for a in "10 2048 lv" "2 2048 lv" "50 2048 lv" "50 16384 lv" "10 2048 kl" "2 2048 kl" "50 2048 kl"; do python tools/bridge_step_profile.py $a; done
  echo "# SDEH_BRIDGE_SEQ=1: the step-sequential forward kernel (lv: fused inference backward; kl: plane backward)"
  for a in "10 2048 lv" "50 2048 lv" "50 16384 lv" "10 2048 kl"; do SDEH_BRIDGE_SEQ=1 python tools/bridge_step_profile.py $a; done
  echo "# SDEH_BWD_PLANES=1: round 3 (step-sequential forward, plane-writing backward)"
  for a in "10 2048 lv" "2 2048 lv" "50 2048 lv" "10 2048 kl"; do SDEH_BWD_PLANES=1 python tools/bridge_step_profile.py $a; done
  for c in "10 2048" "50 16384"; do set -- $c
    (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/bridge_prof -o s$1 -- python $ROOT/tools/bridge_step_profile.py $1 $2 lv > /dev/null 2>&1)
    echo "#"; echo "# per kernel (rocprofv3 --kernel-trace --stats, 6 steps): d = $1, B = $2, lv"
    python tools/rocprof_summary.py /tmp/bridge_prof/s$1_results.db | head -12 | cut -c1-170
  done
  echo "#"
  echo "# matrix-pipe fractions at d = 50, B = 16 384: divergence backward 6 x 64 instructions per (32 rows, coordinate): 102 400 x 50 x 384 x 64 cycles"
  echo "#   / (1024 SIMDs x 2.4 GHz) = 51.2 ms; inference forward pass 2 x 64 per (row tile, coordinate) + ~200 per row tile: 17.6 ms"
} 2>&1 | grep -v amdgpu.ids > $OUT/r04_bridge_step_kernels.txt
# 7. host cost of an evaluation call
{ python tools/eval_host_profile.py; python tools/eval_host_profile.py gmm50_pis_headline 1024; } 2>&1 | grep -v amdgpu.ids > $OUT/r04_eval_host_profile.txt
# 8. the suite and the smoke test
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 > $OUT/r04_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> $OUT/r04_pytest_gpu.txt
cat $OUT/r04_pytest_gpu.txt; head -5 $OUT/r04_kernel_stats_headline.txt | cut -c1-170; tail -c 600 $OUT/r04_bench_headline.json; echo; cat $OUT/r04_training_backward_roofline.txt | cut -c1-200
