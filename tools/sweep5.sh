cd /root/repo; mkdir -p gpurun_out
for F in 0 524288 1048576 1572864; do
for rep in 1 2; do
HRX_DEBUG_FLAGS=$F python bench.py --steps 20 --warmup 3 --config regex23 --batch 262144 --len 2047 --rows 2048 --no-cpu-baseline > gpurun_out/s5_r23_${F}_$rep.json 2>> gpurun_out/s5.err
HRX_DEBUG_FLAGS=$F python bench.py --steps 20 --warmup 3 --config headers3 --batch 65536 --len 2047 --rows 2048 --no-cpu-baseline > gpurun_out/s5_headers3_2k_${F}_$rep.json 2>> gpurun_out/s5.err
done
done
grep -v amdgpu.ids gpurun_out/s5.err | tail -5
