#!/bin/bash
# Kernel timelines of one REPLAYED optimisation step (tools/train_demo.py --graph under rocprofv3 --kernel-trace) at the reference's
# training batches, BASELINE configs[1] / configs[2] with method kl:   bash tools/graph_step_trace.sh [tag]  ->  gpurun_out/step_trace_<tag>.txt
ROOT=$(pwd)
TAG=${1:-now}
OUT=$ROOT/gpurun_out/step_trace_$TAG.txt
: > $OUT
for c in "cfg2_gmm2_dis_kl 512" "cfg2_gmm2_dis_kl 2048" "cfg3_gmm50_pis_kl 512" "cfg3_gmm50_pis_kl 2048"; do
  set -- $c
  D=$ROOT/gpurun_out/steptrace_tmp
  rm -rf $D
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d $D -- python $ROOT/tools/train_demo.py $1 --method kl --batch $2 --steps 60 --seed 1 --graph > $D.log 2>&1)
  DB=$(find $D -name "*.db" | head -1)
  echo "== $1 kl B=$2 (replayed hipGraph step)" >> $OUT
  tail -2 $D.log >> $OUT
  python $ROOT/tools/graph_step_trace.py $DB >> $OUT
  rm -rf $D
done
grep -E "^==|^# one|^# busy" $OUT
