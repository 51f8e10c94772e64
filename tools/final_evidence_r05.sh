cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests -q -m gpu 2>&1 | tail -3 > gpurun_out/r05_gpu_tests.txt; cat gpurun_out/r05_gpu_tests.txt
(python3 tools/dn_bench.py 65536 2048; python3 tools/dn_bench.py 65536 1024; python3 tools/dn_bench.py 262144 1024) 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_dn_bench.txt; tail -4 gpurun_out/r05_dn_bench.txt
bash tools/config_sweep.sh 2>&1 | grep -v "amdgpu.ids\|^Search\|^HIP kernel\|^For debugging\|^Compile\|graph capture\|eager launches" | tail -24
