set -u
mkdir -p gpurun_out/v8
python bench.py > gpurun_out/v8/bench.json 2> gpurun_out/v8/bench.err
tail -c 1500 gpurun_out/v8/bench.json
ROOT=$(pwd)
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/v8/prof -- python $ROOT/bench.py --no-cpu-baseline > $ROOT/gpurun_out/v8/prof_bench.json 2> $ROOT/gpurun_out/v8/prof.err)
DB=$(find gpurun_out/v8/prof -name "*.db" | head -1)
python tools/rocprof_summary.py $DB > gpurun_out/v8/kernel_stats.txt
cat gpurun_out/v8/prof_bench.json >> gpurun_out/v8/kernel_stats.txt
find gpurun_out/v8/prof -name "*.db" -delete
head -12 gpurun_out/v8/kernel_stats.txt
bash tools/pmc_profile.sh gpurun_out/v8/pmc > /dev/null 2>&1
cat gpurun_out/v8/pmc/summary.txt | head -50
python tools/all_configs_timing.py 2>&1 | tee gpurun_out/v8/all_configs.txt
python tests/perf/integrator_timing.py 2>&1 | tee gpurun_out/v8/integrator.txt
