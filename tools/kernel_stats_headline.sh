OUT=gpurun_out/r02g; mkdir -p $OUT; ROOT=$(pwd)
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_headline -- python $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra > $ROOT/$OUT/prof_headline.log 2>&1)
DB=$(find $OUT/prof_headline -name "*.db" | head -1)
python tools/rocprof_summary.py $DB > $OUT/kernel_stats_headline.txt
tail -1 $OUT/prof_headline.log | cut -c1-600 >> $OUT/kernel_stats_headline.txt
find $OUT/prof_headline -name "*.db" -delete
head -6 $OUT/kernel_stats_headline.txt | cut -c1-160
