for w in cfg5_like_bridge196 wide_pis_funnel196; do
python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'])"
done
