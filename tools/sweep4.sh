cd /root/repo; mkdir -p gpurun_out
for F in 0 524288; do
HRX_DEBUG_FLAGS=$F python bench.py --steps 20 --warmup 3 --config regex23 --batch 262144 --len 2047 --rows 2048 --no-cpu-baseline > gpurun_out/s4_r23_$F.json 2>> gpurun_out/s4.err
HRX_DEBUG_FLAGS=$F python bench.py --steps 5 --warmup 2 --config headers3 --batch 32768 --len 32767 --rows 32768 --no-cpu-baseline > gpurun_out/s4_headers3_full_$F.json 2>> gpurun_out/s4.err
HRX_DEBUG_FLAGS=$F python bench.py --steps 20 --warmup 3 --config headers3 --batch 65536 --len 2047 --rows 2048 --no-cpu-baseline > gpurun_out/s4_headers3_2k_$F.json 2>> gpurun_out/s4.err
HRX_DEBUG_FLAGS=$F python bench.py --steps 20 --warmup 3 --config regex123 --batch 65536 --len 1023 --rows 1024 --no-cpu-baseline > gpurun_out/s4_r123_$F.json 2>> gpurun_out/s4.err
HRX_DEBUG_FLAGS=$((F+3)) python bench.py --steps 20 --warmup 3 --config headers3 --batch 65536 --len 2047 --rows 2048 --no-cpu-baseline > gpurun_out/s4_headers3_2k_nostore_$F.json 2>> gpurun_out/s4.err
HRX_DEBUG_FLAGS=$((F+3)) python bench.py --steps 20 --warmup 3 --config regex23 --batch 262144 --len 2047 --rows 2048 --no-cpu-baseline > gpurun_out/s4_r23_nostore_$F.json 2>> gpurun_out/s4.err
HRX_DEBUG_FLAGS=$((F+3)) python bench.py --steps 50 --warmup 3 --no-cpu-baseline > gpurun_out/s4_default_nostore_$F.json 2>> gpurun_out/s4.err
done
grep -v amdgpu.ids gpurun_out/s4.err | tail -5
