#!/bin/bash
# Round-3 measurement set (run on the GPU box from the repo root): bash tools/measure_r03.sh gpurun_out/r03final
set -u
OUT=${1:-gpurun_out/r03final}
mkdir -p $OUT
ROOT=$(pwd)
python bench.py > $OUT/bench_headline.json 2> $OUT/bench_headline.err; tail -c 300 $OUT/bench_headline.json; echo
python bench.py --workload wide_pis_funnel196 > $OUT/bench_wide_pis.json 2>/dev/null
python bench.py --workload cfg5_like_bridge196 > $OUT/bench_cfg5_like.json 2>/dev/null
python bench.py --workload train_cfg5_like > $OUT/bench_train_cfg5_like.json 2>/dev/null
python bench.py --workload train_wide_pis_lv > $OUT/bench_train_wide_pis_lv.json 2>/dev/null
python bench.py --workload train_gmm2_dis_kl > $OUT/bench_train_gmm2_dis_kl.json 2>/dev/null
python bench.py --workload train_gmm50_pis_kl --steps 40 > $OUT/bench_train_gmm50_pis_kl.json 2>/dev/null
stats() {  # name, command...
  local name=$1; shift
  (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_$name -- "$@" > $ROOT/$OUT/prof_$name.log 2>&1)
  local DB=$(find $OUT/prof_$name -name "*.db" | head -1)
  python tools/rocprof_summary.py $DB > $OUT/kernel_stats_$name.txt
  tail -2 $OUT/prof_$name.log | cut -c1-700 >> $OUT/kernel_stats_$name.txt
  find $OUT/prof_$name -name "*.db" -delete
  head -6 $OUT/kernel_stats_$name.txt | cut -c1-170
}
stats headline python $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra
stats train_cfg5_like python $ROOT/tools/wide_train_timing.py cfg5_like_bridge196 4096 lv
stats train_wide_pis_lv python $ROOT/tools/wide_train_timing.py wide_pis_funnel196 8192 lv
stats train_wide_pis_kl python $ROOT/tools/wide_train_timing.py wide_pis_funnel196 8192 kl
(for c in "cfg3_gmm50_pis_kl kl" "cfg2_gmm2_dis_kl kl" "cfg1_dw_dis_lv lv" "cfg4_funnel_dds_lv lv"; do python tools/bwd_timing.py $c 2048 32768 65536; done) 2>&1 | grep -v amdgpu > $OUT/bwd_timing.txt
python tools/mid_batch_timing.py 2>&1 | grep -v amdgpu > $OUT/mid_batch.txt
python -m pytest tests -q -m gpu -n 4 2>&1 | tail -3 > $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> $OUT/pytest_gpu.txt
cat $OUT/bwd_timing.txt | cut -c1-200; cat $OUT/pytest_gpu.txt
