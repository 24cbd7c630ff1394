ROOT=$(pwd)
mkdir -p gpurun_out/trainprof
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/trainprof/p -- python $ROOT/tools/train_demo.py ${NAME:-cfg1_dw_dis_lv} --steps 200 --seed 1 ${EXTRA:-} > $ROOT/gpurun_out/trainprof/out.txt 2>&1)
DB=$(find gpurun_out/trainprof/p -name "*.db" | head -1)
python tools/rocprof_summary.py $DB | head -24
find gpurun_out/trainprof/p -name "*.db" -delete
tail -4 gpurun_out/trainprof/out.txt
