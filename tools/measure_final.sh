OUT=gpurun_out/r02f; mkdir -p $OUT
python bench.py > $OUT/bench_headline.json 2> $OUT/bench_headline.err; tail -c 200 $OUT/bench_headline.json; echo
python bench.py --workload train_gmm2_dis_kl > $OUT/bench_train_gmm2_dis_kl.json 2>/dev/null
python bench.py --workload train_gmm50_pis_kl --steps 40 > $OUT/bench_train_gmm50_pis_kl.json 2>/dev/null
(for c in "cfg3_gmm50_pis_kl kl" "cfg2_gmm2_dis_kl kl" "cfg1_dw_dis_lv lv" "cfg4_funnel_dds_lv lv"; do python tools/bwd_timing.py $c 2048 32768 65536; done) 2>&1 | grep -v amdgpu > $OUT/bwd_timing.txt
bash tools/train_profile_set.sh $OUT/trainset > /dev/null 2>&1; cp $OUT/trainset/summary.txt $OUT/train_kernel_stats.txt
(for a in "cfg2_gmm2_dis_kl --batch 2048 --steps 300" "cfg2_gmm2_dis_kl --batch 65536 --steps 100" "cfg3_gmm50_pis_kl --batch 2048 --steps 200" "cfg3_gmm50_pis_kl --batch 65536 --steps 100" "cfg1_dw_dis_lv --batch 2048 --steps 300" "cfg1_dw_dis_lv --batch 65536 --steps 100"; do echo "## tools/train_demo.py $a --seed 1 --graph"; python tools/train_demo.py $a --seed 1 --graph 2>&1 | grep -E "ms/step" | tail -1; done) > $OUT/train_graph_timing.txt 2>&1
python tools/small_batch_timing.py 2>&1 | grep -v amdgpu > $OUT/small_batch.txt; python tools/mid_batch_timing.py 2>&1 | grep -v amdgpu > $OUT/mid_batch.txt
cat $OUT/bwd_timing.txt | cut -c1-170; cat $OUT/train_graph_timing.txt
