#!/bin/bash
# rocprofv3 kernel-trace summary of training steps on the wide networks (run on the GPU box from the repo root):
#   bash tools/wide_train_profile.sh <out dir> <spec> <batch> <method> [steps]
set -u
OUT=${1:-gpurun_out/wide_train_prof}; SPEC=${2:-wide_pis_funnel196}; B=${3:-8192}; M=${4:-lv}; T=${5:-}
ROOT=$(pwd)
mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -- python $ROOT/tools/wide_train_timing.py $SPEC $B $M $T > $ROOT/$OUT/run.log 2>&1)
DB=$(find $OUT/prof -name "*.db" | head -1)
python tools/rocprof_summary.py $DB > $OUT/kernel_stats_${SPEC}_${M}_b${B}.txt
tail -4 $OUT/run.log >> $OUT/kernel_stats_${SPEC}_${M}_b${B}.txt
find $OUT/prof -name "*.db" -delete
cat $OUT/kernel_stats_${SPEC}_${M}_b${B}.txt
