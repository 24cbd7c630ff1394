# Steady-state wall time per optimisation step (sample -> loss -> backward -> clip -> Adam), eager launches against one hipGraph
# replay per step (sde_sampler_amd/utils/graphs.py).  B = 2048, T = 100.  Usage: bash tools/train_timing.sh > profiles/...
for cfg in "cfg1_dw_dis_lv" "cfg2_gmm2_dis_kl" "bridge_dw --lr 2e-3"; do
  for mode in "" "--graph"; do
    echo "## tools/train_demo.py $cfg --steps 300 --seed 1 $mode"
    python tools/train_demo.py $cfg --steps 300 --seed 1 $mode 2>&1 | grep -E "^step 300|RESULT"
  done
done
