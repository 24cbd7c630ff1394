#!/bin/bash
# Per-kernel time of a set of training loops (run on the GPU box from the repo root): NAME:BATCH:STEPS triples.
#   bash tools/train_profile_set.sh gpurun_out/trainset "cfg2_gmm2_dis_kl:2048:100 cfg2_gmm2_dis_kl:65536:30"
OUT=${1:-gpurun_out/trainset}
SET=${2:-"cfg2_gmm2_dis_kl:2048:100 cfg2_gmm2_dis_kl:65536:30 cfg3_gmm50_pis_kl:2048:100 cfg3_gmm50_pis_kl:65536:20 cfg1_dw_dis_lv:2048:100 cfg1_dw_dis_lv:65536:30"}
ROOT=$(pwd)
mkdir -p $OUT
for item in $SET; do
  IFS=: read name batch steps <<< "$item"
  tag=${name}_${batch}
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/p_$tag -- python $ROOT/tools/train_demo.py $name --batch $batch --steps $steps --seed 1 ${EXTRA:-} > $ROOT/$OUT/out_$tag.txt 2>&1)
  DB=$(find $OUT/p_$tag -name "*.db" | head -1)
  echo "## train_demo.py $name --batch $batch --steps $steps ${EXTRA:-}" | tee -a $OUT/summary.txt
  python tools/rocprof_summary.py $DB | head -12 | cut -c1-200 | tee -a $OUT/summary.txt
  grep -E "ms/step" $OUT/out_$tag.txt | tail -1 | tee -a $OUT/summary.txt
  find $OUT/p_$tag -name "*.db" -delete
done
