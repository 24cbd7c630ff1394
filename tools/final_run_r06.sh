#!/bin/bash
# Round-6 measurement run (on the GPU box, from the repo root): everything whose summary is committed under profiles/r06_*.
#   gpurun --timeout 3000 -- 'bash tools/final_run_r06.sh [sections]'      sections (default: all): pmc stats bench nice train configs tests
set -u
OUT=gpurun_out/r06
mkdir -p $OUT
ROOT=$(pwd)
WANT=${*:-pmc stats bench nice train configs tests}
has() { [[ " $WANT " == *" $1 "* ]]; }
if has pmc; then
  # PMC passes of the headline trajectory kernel (separate --pmc passes, kernel trace only) + the stamped record bench.py reads
  bash tools/pmc_profile.sh $OUT/pmc_headline > $OUT/pmc.log 2>&1
  { echo "# rocprofv3 PMC passes (tools/pmc_profile.sh, separate --pmc passes with --kernel-trace only) of the headline trajectory kernel traj_ws<50_0_pis_gmm4>,"
    echo "# GMM-40 d=50, B=65536, T=100, round 6; per launch, averaged over the dispatches.  GRBM_GUI_ACTIVE is summed over the 8 XCDs."
    cat $OUT/pmc_headline/summary.txt; } > $OUT/r06_pmc_headline.txt
  cp $OUT/r06_pmc_headline.txt profiles/r06_pmc_headline.txt
  python tools/pmc_headline_json.py $OUT/pmc_headline/summary.txt profiles/r06_pmc_headline.txt > /dev/null; cp profiles/pmc_headline.json $OUT/
fi
if has stats; then
  # rocprofv3 kernel trace of the bench command
  (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_headline -- python $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra > $ROOT/$OUT/prof_headline.log 2>&1)
  DB=$(find $OUT/prof_headline -name "*.db" | head -1)
  python tools/rocprof_summary.py $DB > $OUT/r06_kernel_stats_headline.txt
  tail -2 $OUT/prof_headline.log | cut -c1-900 >> $OUT/r06_kernel_stats_headline.txt
  find $OUT/prof_headline -name "*.db" -delete
fi
if has bench; then
  python bench.py > $OUT/r06_bench_headline.json 2> $OUT/bench_headline.err
  python bench.py --dist --no-extra --no-cpu-baseline > $OUT/r06_bench_headline_dist1.json 2>> $OUT/bench_headline.err
fi
if has nice; then
  # BASELINE configs[4] as written (target = nice): evaluation and one optimisation step; kernel statistics of the evaluation
  python bench.py --workload cfg5_nice_bridge196 > $OUT/r06_bench_cfg5_nice_bridge196.json 2> $OUT/bench_nice.err
  python bench.py --workload cfg5_like_bridge196 --no-cpu-baseline > $OUT/r06_bench_cfg5_like_bridge196.json 2>> $OUT/bench_nice.err
  python bench.py --workload train_cfg5_nice --no-cpu-baseline > $OUT/r06_bench_train_cfg5_nice.json 2>> $OUT/bench_nice.err
  (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_nice -- python $ROOT/bench.py --workload cfg5_nice_bridge196 --steps 2 --warmup 1 --no-cpu-baseline > $ROOT/$OUT/prof_nice.log 2>&1)
  DB=$(find $OUT/prof_nice -name "*.db" | head -1)
  python tools/rocprof_summary.py $DB > $OUT/r06_kernel_stats_cfg5_nice_bridge196.txt
  find $OUT/prof_nice -name "*.db" -delete
  python tools/nice_timing.py 2>&1 | grep -v amdgpu.ids > $OUT/r06_nice_score_timing.txt
fi
if has train; then
  for w in train_gmm2_dis_kl train_gmm50_pis_kl; do python bench.py --workload $w --no-cpu-baseline > $OUT/r06_bench_$w.json 2>> $OUT/bench_train.err; done
fi
if has configs; then
  python tools/all_configs_timing.py 2>&1 | grep -v amdgpu.ids > $OUT/r06_all_configs_timing.txt
fi
if has tests; then
  rm -f gpurun_out/parity_measured.txt gpurun_out/fuzz_hatches.txt
  { python -m pytest tests -q -m gpu 2>&1 | tail -4; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids; } > $OUT/r06_pytest_gpu.txt
  sort gpurun_out/parity_measured.txt > $OUT/r06_parity_measured.txt
fi
ls $OUT
