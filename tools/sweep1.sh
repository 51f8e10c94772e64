# config sweep behind DESIGN.md's per-config table: run on the GPU box via gpurun
set -x
cd /root/repo; mkdir -p gpurun_out
python bench.py --steps 100 --warmup 10 > gpurun_out/s1_default.json 2> gpurun_out/s1_default.err
for L in position-major string-major; do
python bench.py --steps 20 --warmup 3 --config regex23 --batch 262144 --len 2047 --rows 2048 --layout $L --no-cpu-baseline > gpurun_out/s1_r23_$L.json 2>> gpurun_out/s1.err
python bench.py --steps 20 --warmup 3 --config regex123 --batch 65536 --len 1023 --rows 1024 --layout $L --no-cpu-baseline > gpurun_out/s1_r123_$L.json 2>> gpurun_out/s1.err
python bench.py --steps 20 --warmup 3 --config dfa256 --batch 65536 --len 4095 --rows 4096 --layout $L --no-cpu-baseline > gpurun_out/s1_dfa256_$L.json 2>> gpurun_out/s1.err
python bench.py --steps 20 --warmup 3 --config regex1 --batch 8192 --len 32767 --rows 32768 --layout $L --no-cpu-baseline > gpurun_out/s1_long_$L.json 2>> gpurun_out/s1.err
done
python bench.py --steps 10 --warmup 2 --config regex23 --batch 1048576 --len 2047 --rows 2048 --no-cpu-baseline > gpurun_out/s1_r23_full.json 2>> gpurun_out/s1.err
python bench.py --steps 10 --warmup 2 --config regex123 --batch 8192 --len 32767 --rows 32768 --no-cpu-baseline > gpurun_out/s1_r123_long.json 2>> gpurun_out/s1.err
tail -5 gpurun_out/s1.err
