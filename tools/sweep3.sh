set -x
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/s3_pytest.txt
cat gpurun_out/s3_pytest.txt
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/s3_default.json 2> gpurun_out/s3.err
HRX_DEBUG_FLAGS=524288 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/s3_default_narrow.json 2>> gpurun_out/s3.err
python bench.py --steps 20 --warmup 3 --config regex23 --batch 262144 --len 2047 --rows 2048 --no-cpu-baseline > gpurun_out/s3_r23.json 2>> gpurun_out/s3.err
python bench.py --steps 5 --warmup 2 --config headers3 --batch 32768 --len 32767 --rows 32768 --no-cpu-baseline > gpurun_out/s3_headers3_full.json 2>> gpurun_out/s3.err
python bench.py --steps 20 --warmup 3 --config headers3 --batch 65536 --len 2047 --rows 2048 --no-cpu-baseline > gpurun_out/s3_headers3_2k.json 2>> gpurun_out/s3.err
python bench.py --steps 20 --warmup 3 --config regex1 --batch 8192 --len 32767 --rows 32768 --no-cpu-baseline > gpurun_out/s3_long.json 2>> gpurun_out/s3.err
python bench.py --steps 50 --warmup 5 --config regex1 --batch 4096 --len 1023 --rows 1024 --no-cpu-baseline > gpurun_out/s3_small.json 2>> gpurun_out/s3.err
grep -v amdgpu.ids gpurun_out/s3.err | tail -5
