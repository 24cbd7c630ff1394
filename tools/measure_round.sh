#!/bin/bash
# Round-end measurement set (run on the GPU box from the repo root): bench lines of the three workloads, rocprofv3 kernel stats
# and PMC passes.  Usage: bash tools/measure_round.sh gpurun_out/r02
set -u
OUT=${1:-gpurun_out/round}
mkdir -p $OUT
ROOT=$(pwd)
python bench.py > $OUT/bench_headline.json 2> $OUT/bench_headline.err; tail -c 600 $OUT/bench_headline.json; echo
python bench.py --workload wide_pis_funnel196 > $OUT/bench_wide_pis.json 2> $OUT/bench_wide_pis.err; tail -c 400 $OUT/bench_wide_pis.json; echo
python bench.py --workload cfg5_like_bridge196 > $OUT/bench_cfg5_like.json 2> $OUT/bench_cfg5_like.err; tail -c 400 $OUT/bench_cfg5_like.json; echo
stats() {  # name, command...
  local name=$1; shift
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_$name -- "$@" > $ROOT/$OUT/prof_$name.log 2>&1)
  local DB=$(find $OUT/prof_$name -name "*.db" | head -1)
  python tools/rocprof_summary.py $DB > $OUT/kernel_stats_$name.txt
  tail -3 $OUT/prof_$name.log >> $OUT/kernel_stats_$name.txt
  find $OUT/prof_$name -name "*.db" -delete
  head -8 $OUT/kernel_stats_$name.txt
}
stats headline python $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra
stats wide_pis python $ROOT/tools/wide_timing.py wide_pis_funnel196 32768
stats cfg5_like python $ROOT/tools/wide_timing.py cfg5_like_bridge196 4096
bash tools/pmc_profile.sh $OUT/pmc_headline > /dev/null 2>&1; cp $OUT/pmc_headline/summary.txt $OUT/pmc_summary_headline.txt
bash tools/pmc_profile.sh $OUT/pmc_wide_pis "python tools/wide_timing.py wide_pis_funnel196 32768" > /dev/null 2>&1; cp $OUT/pmc_wide_pis/summary.txt $OUT/pmc_summary_wide_pis.txt
EM_STEPS=40 bash tools/pmc_profile.sh $OUT/pmc_cfg5_like "python tools/wide_timing.py cfg5_like_bridge196 4096" > /dev/null 2>&1; cp $OUT/pmc_cfg5_like/summary.txt $OUT/pmc_summary_cfg5_like.txt
head -40 $OUT/pmc_summary_cfg5_like.txt
