set -u
mkdir -p gpurun_out/exp4
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "dense_mixture or mixture_table or padded_reference" 2>&1 | tail -8 > gpurun_out/exp4/pytest.txt
python tools/dense_mixture_timing.py > gpurun_out/exp4/mm.txt 2>&1
cat gpurun_out/exp4/pytest.txt; grep -v "amdgpu.ids" gpurun_out/exp4/mm.txt; grep dense_mixture gpurun_out/parity_measured.txt
