set -x
cd /root/repo; mkdir -p gpurun_out
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/s2_pytest.txt
python bench.py --steps 5 --warmup 2 --config headers3 --batch 32768 --len 32767 --rows 32768 --no-cpu-baseline > gpurun_out/s2_headers3_full.json 2> gpurun_out/s2.err
python bench.py --steps 20 --warmup 3 --config headers3 --batch 65536 --len 2047 --rows 2048 --no-cpu-baseline > gpurun_out/s2_headers3_2k.json 2>> gpurun_out/s2.err
cat gpurun_out/s2_pytest.txt; grep -v amdgpu.ids gpurun_out/s2.err | tail -5
