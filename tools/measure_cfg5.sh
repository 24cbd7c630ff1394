OUT=gpurun_out/r02e; mkdir -p $OUT; ROOT=$(pwd)
python bench.py --workload cfg5_like_bridge196 > $OUT/bench_cfg5_like.json 2> $OUT/bench_cfg5_like.err; tail -c 300 $OUT/bench_cfg5_like.json; echo
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_cfg5 -- python $ROOT/tools/wide_timing.py cfg5_like_bridge196 4096 > $ROOT/$OUT/prof_cfg5.log 2>&1)
DB=$(find $OUT/prof_cfg5 -name "*.db" | head -1); python tools/rocprof_summary.py $DB > $OUT/kernel_stats_cfg5_like.txt; tail -3 $OUT/prof_cfg5.log >> $OUT/kernel_stats_cfg5_like.txt; find $OUT/prof_cfg5 -name "*.db" -delete; head -6 $OUT/kernel_stats_cfg5_like.txt
EM_STEPS=40 bash tools/pmc_profile.sh $OUT/pmc_cfg5_like "python tools/wide_timing.py cfg5_like_bridge196 4096" > /dev/null 2>&1; cp $OUT/pmc_cfg5_like/summary.txt $OUT/pmc_summary_cfg5_like.txt; grep -E "MFMA|GRBM_GUI|INSTS_VALU " $OUT/pmc_summary_cfg5_like.txt
