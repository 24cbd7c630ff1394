#!/bin/bash
# Whole optimisation steps (one hipGraph per step, tools/train_demo.py --graph) at the reference's training batch sizes, back-propagation
# through time: the 16-trajectory backward (csrc/sdeh_bwdf16.hip, the default up to 8192 trajectories) against SDEH_BWD_TILE=32.
for item in cfg2_gmm2_dis_kl:512:300 cfg2_gmm2_dis_kl:2048:300 cfg3_gmm50_pis_kl:512:200 cfg3_gmm50_pis_kl:2048:200; do
  IFS=: read name batch steps <<< "$item"
  for tile in 32 16; do
    echo "## SDEH_BWD_TILE=$tile tools/train_demo.py $name --batch $batch --steps $steps --seed 1 --graph"
    SDEH_BWD_TILE=$tile python tools/train_demo.py $name --batch $batch --steps $steps --seed 1 --graph 2>&1 | grep "ms/step" | tail -1
  done
done
