#!/bin/bash
# Round-5 measurement run (on the GPU box, from the repo root): everything whose summary is committed under profiles/r05_*.
#   gpurun --timeout 3000 -- 'bash tools/final_run_r05.sh'
set -u
OUT=gpurun_out/r05
mkdir -p $OUT
ROOT=$(pwd)
# 1. PMC passes of the headline trajectory kernel (separate --pmc passes, kernel trace only) + the stamped record bench.py reads
bash tools/pmc_profile.sh $OUT/pmc_headline > $OUT/pmc.log 2>&1
{ echo "# rocprofv3 PMC passes (tools/pmc_profile.sh, separate --pmc passes with --kernel-trace only) of the headline trajectory kernel traj_ws<50_0_pis_gmm4>,"
  echo "# GMM-40 d=50, B=65536, T=100, round 5; per launch, averaged over the dispatches.  GRBM_GUI_ACTIVE is summed over the 8 XCDs."
  cat $OUT/pmc_headline/summary.txt; } > $OUT/r05_pmc_headline.txt
cp $OUT/r05_pmc_headline.txt profiles/r05_pmc_headline.txt
python tools/pmc_headline_json.py $OUT/pmc_headline/summary.txt profiles/r05_pmc_headline.txt > /dev/null; cp profiles/pmc_headline.json $OUT/
# 2. rocprofv3 kernel trace of the bench command
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_headline -- python $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra > $ROOT/$OUT/prof_headline.log 2>&1)
DB=$(find $OUT/prof_headline -name "*.db" | head -1)
python tools/rocprof_summary.py $DB > $OUT/r05_kernel_stats_headline.txt
tail -2 $OUT/prof_headline.log | cut -c1-900 >> $OUT/r05_kernel_stats_headline.txt
find $OUT/prof_headline -name "*.db" -delete
# 3. the bench lines
python bench.py > $OUT/r05_bench_headline.json 2> $OUT/bench_headline.err
python bench.py --dist --no-extra --no-cpu-baseline > $OUT/r05_bench_headline_dist1.json 2>> $OUT/bench_headline.err
for w in train_gmm2_dis_kl train_gmm50_pis_kl; do python bench.py --workload $w --no-cpu-baseline > $OUT/r05_bench_$w.json 2>> $OUT/bench_train.err; done
# 4. every BASELINE configuration at its per-GPU batch; training forward + backward with the pre-activation record against the re-evaluating launches
python tools/all_configs_timing.py 2>&1 | grep -v amdgpu.ids > $OUT/r05_all_configs_timing.txt
{ echo "# tools/zrec_ab.py: training forward + fused backward kernel times (HIP events) WITH the pre-activation record (ABI v6, default) against the"
  echo "# re-evaluating launches (plan option SDEH_BWD_ZREC=0), identical Philox draws; gradient difference of the two relative to each tensor's scale.  MI355X, round 5."
  REPS=7 python tools/zrec_ab.py 2>&1 | grep -v "Warn\|loss0\|Consider"
  echo "# small batches through time (16-trajectory teams, scan form) and mid batches"
  REPS=7 python tools/zrec_ab.py cfg3_gmm50_pis_kl:kl:2048 cfg3_gmm50_pis_kl:kl:8192 cfg2_gmm2_dis_kl:kl:512 cfg2_gmm2_dis_kl:kl:2048 cfg1_dw_dis_lv:kl:2048 cfg4_funnel_dds_lv:kl:2048 cfg3_gmm50_pis_kl:kl:32768 cfg4_funnel_dds_lv:kl:16384 2>&1 | grep -v "Warn\|loss0\|Consider"
  echo "# trajectory-split teams forced for two coordinate tiles through time (plan option SDEH_BWD_V2)"
  SDEH_BWD_V2=1 REPS=5 python tools/zrec_ab.py cfg3_gmm50_pis_kl:kl:65536 2>&1 | grep -v "Warn\|loss0\|Consider"; } 2>&1 | grep -v amdgpu.ids > $OUT/r05_training_backward_roofline.txt
# 5. phase profiles (measurement builds under prof_tmp/): fused backward with the record; forward cost of the record stores
[ -f prof_tmp/libsdeh_prof.so ] && bash tools/bwdf_phase_profile.sh run 2>&1 | grep -v amdgpu.ids > $OUT/r05_bwd_fused_phases.txt
[ -f prof_tmp/libsdeh_zabl_nt.so ] && bash tools/zrec_fwd_ablation.sh run 2>&1 | grep -v "amdgpu.ids\|phases" > $OUT/r05_zrec_fwd_ablation.txt
# 6. PMC passes of the fused backward with / without the record; kernel trace of a training bench
bash tools/pmc_bwd.sh > $OUT/pmc_bwd.log 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_train -- python $ROOT/bench.py --workload train_gmm2_dis_kl --steps 10 --warmup 3 --no-cpu-baseline > $ROOT/$OUT/prof_train.log 2>&1)
DB=$(find $OUT/prof_train -name "*.db" | head -1)
python tools/rocprof_summary.py $DB > $OUT/r05_kernel_stats_train_gmm2_dis_kl.txt
find $OUT/prof_train -name "*.db" -delete
# 7. replayed optimisation steps at the reference's training batches: kernel timelines + whole-step times
bash tools/graph_step_trace.sh r05 > /dev/null 2>&1; cp gpurun_out/step_trace_r05.txt $OUT/r05_graph_step_trace.txt
python tools/train_reference_schedule.py --steps 2000 --out /tmp/ref_schedule_tmp.pt 2>&1 | grep -v amdgpu.ids | tail -12 > $OUT/r05_train_reference_schedule.txt
# 8. Bridge training steps (unchanged kernels except the generative network's record) and the quality runs
{ echo "# Bridge training step (loss + backward), T = 200, wall clock, eager, best of 5 (tools/bridge_step_profile.py <d> <B> <method>); MI355X, round 5"
  for a in "10 2048 lv" "2 2048 lv" "50 2048 lv" "50 16384 lv" "10 2048 kl"; do python tools/bridge_step_profile.py $a; done; } 2>&1 | grep -v amdgpu.ids > $OUT/r05_bridge_step_kernels.txt
{ for a in "cfg1_dw_dis_lv --steps 10000 --lr 1e-3 --eval-batch 262144" "cfg1_dw_dis_lv --method kl --steps 10000 --lr 1e-3 --eval-batch 262144" "cfg2_gmm2_dis_kl --method lv --steps 6000 --lr 1e-3" "bridge_dw --steps 400"; do
    echo "## tools/train_demo.py $a --graph --seed 1"; python tools/train_demo.py $a --graph --seed 1 2>&1 | grep -E "^\[|RESULT|step (400|6000|10000):"; done; } > $OUT/r05_train_quality.txt 2>&1
# 9. host cost of an evaluation call
{ python tools/eval_host_profile.py; python tools/eval_host_profile.py gmm50_pis_headline 1024; } 2>&1 | grep -v amdgpu.ids > $OUT/r05_eval_host_profile.txt
# 9b. dense mixtures: matrix-pipe contractions (default) / exact form with scalar-cache tables (SDEH_GMM_MM=0) / round 4's LDS tables (measurement
#     build prof_tmp/libsdeh_gmmlds.so = -DSDEH_GMM_SGPR=0); PMC passes of the dense shared-scale kernel in both forms; the micro-benchmarks
{ echo "# tools/dense_mixture_timing.py on the MI355X, round 5: B = 65 536, T = 100, kernel ms over 7 launches (HIP events); x_T hashes: identical Philox draws"
  echo "## default: both mixture contractions on the matrix pipe where the binding vouches for the product form (kernel names ...,mm)"
  python tools/dense_mixture_timing.py
  echo "## plan option SDEH_GMM_MM=0: exact form, tables through the scalar cache (gmm_online_s)"
  SDEH_GMM_MM=0 python tools/dense_mixture_timing.py
  if [ -f prof_tmp/libsdeh_gmmlds.so ]; then
    echo "## exact form, tables as LDS broadcast reads (-DSDEH_GMM_SGPR=0 build = round 4's path)"
    SDEH_GMM_MM=0 SDEH_LIBRARY=$ROOT/prof_tmp/libsdeh_gmmlds.so python tools/dense_mixture_timing.py
  fi; } 2>&1 | grep -v "amdgpu.ids\|^# library" > $OUT/r05_dense_mixture_timing.txt
for form in mm:"" valu:0; do
  tag=${form%%:*}; v=${form#*:}
  SDEH_GMM_MM=$v bash tools/pmc_profile.sh $OUT/pmc_dense_$tag "python bench.py --workload gmm50_dense_shared --steps 4 --warmup 1 --no-cpu-baseline --no-extra --no-graphed" > /dev/null 2>&1
  { echo "# PMC passes (tools/pmc_profile.sh) of traj_ws<50_0_pis_gmm> on gmm50_dense_shared, mixture form: $tag (mm = matrix pipe, valu = exact form on the vector pipe, scalar-cache tables); per launch"
    cat $OUT/pmc_dense_$tag/summary.txt; } > $OUT/r05_pmc_dense_shared_$tag.txt
done
for u in valu_dep mfma4x4 smem_stream; do [ -x tools/ubench/$u ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/$u.hip -o tools/ubench/$u 2> /dev/null; done
{ echo "# tools/ubench/{valu_dep,mfma4x4,smem_stream}.hip on the MI355X, round 5 (cycles at an ASSUMED 2.4 GHz: ratios are what counts)"
  echo "## valu_dep: issue cost of packed / plain fp32 vector instructions, one and two waves per SIMD"; ./tools/ubench/valu_dep
  echo; echo "## mfma4x4: the mixture's two contractions as v_mfma_f32_4x4x1_16b_f32 in the T layout (layout check against float64 + timing)"; ./tools/ubench/mfma4x4
  echo; echo "## smem_stream: wave-uniform tables as scalar loads / LDS broadcast reads feeding 32 packed instructions per 32-dword batch"
  ./tools/ubench/smem_stream | grep -E "workgroups|table  2 KB|table 16 KB|table 64 KB" | sed 's/(wave 0; memtime ticks x 21 if 100 MHz), //'; } > $OUT/r05_ubench_mixture.txt 2>&1
# 9c. PMC passes of configs[3]'s shard with the out layer as 4 x 4 x 1 row groups / as 32-row tiles
for form in out4:"" tiles32:0; do
  tag=${form%%:*}; v=${form#*:}
  KERNEL="traj_ws_kernel<10," SDEH_WS_OUT4=$v REPS=4 bash tools/pmc_profile.sh $OUT/pmc_cfg4_$tag "python tools/dense_mixture_timing.py" > /dev/null 2>&1
  { echo "# PMC passes (tools/pmc_profile.sh, KERNEL=traj_ws_kernel<10,) of traj_ws<10_0_dds_funnel> on configs[3]'s per-GPU shard (funnel d = 10, B = 32 768, T = 401), out layer: $tag"
    echo "# (out4 = v_mfma_f32_4x4x1 row groups, section 3j; tiles32 = plan option SDEH_WS_OUT4=0: 32-row tiles); per launch"
    cat $OUT/pmc_cfg4_$tag/summary.txt; } > $OUT/r05_pmc_cfg4_shard_$tag.txt
done
# 10. the suite and the smoke test
rm -f gpurun_out/parity_measured.txt gpurun_out/fuzz_hatches.txt
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -4 > $OUT/r05_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> $OUT/r05_pytest_gpu.txt
sort gpurun_out/parity_measured.txt > $OUT/r05_parity_measured.txt
cat $OUT/r05_pytest_gpu.txt; head -5 $OUT/r05_kernel_stats_headline.txt | cut -c1-170; tail -c 900 $OUT/r05_bench_headline.json; echo; cat $OUT/r05_training_backward_roofline.txt | cut -c1-330
