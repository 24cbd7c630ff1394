set -u
OUT=gpurun_out/r03u
mkdir -p $OUT
ROOT=$(pwd)
bash tools/pmc_profile.sh $OUT/pmc_headline > $OUT/pmc.log 2>&1
python tools/pmc_headline_json.py $OUT/pmc_headline/summary.txt profiles/r03_pmc_headline.txt > /dev/null; cp profiles/pmc_headline.json profiles/r03_pmc_headline.txt $OUT/
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_headline -- python $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra > $ROOT/$OUT/prof_headline.log 2>&1)
DB=$(find $OUT/prof_headline -name "*.db" | head -1)
python tools/rocprof_summary.py $DB > $OUT/kernel_stats_headline.txt
tail -2 $OUT/prof_headline.log | cut -c1-700 >> $OUT/kernel_stats_headline.txt
find $OUT/prof_headline -name "*.db" -delete
head -6 $OUT/kernel_stats_headline.txt | cut -c1-170
python bench.py > $OUT/bench_headline.json 2> $OUT/bench_headline.err; tail -c 400 $OUT/bench_headline.json; echo
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 > $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> $OUT/pytest_gpu.txt
cat $OUT/pytest_gpu.txt
