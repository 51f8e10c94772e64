#!/bin/bash
# Round 6: rocprofv3 kernel stats + PMC traffic for BASELINE configs[2], [3] (one GPU's share) and [4] at the sizes of bench.py's other_configs legs — configs[2] and [3] on RECORD PLANES
# (--planes: what the legs run), and configs[3] once more on the interleaved layout for comparison -> gpurun_out/r06_cfg{3,4,4i,5}_{kernel_stats.csv,pmc.json} (copied to profiles/)
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/profile_cfg.sh r06_cfg3 --config regex23 --batch 1048576 --len 2047 --rows 2048 --steps 5 --warmup 2 --distinct 65536 --planes
bash tools/profile_cfg.sh r06_cfg4 --config headers3 --batch 32768 --len 32767 --rows 32768 --steps 5 --warmup 2 --distinct 4096 --planes
bash tools/profile_cfg.sh r06_cfg4i --config headers3 --batch 32768 --len 32767 --rows 32768 --steps 5 --warmup 2 --distinct 4096
bash tools/profile_cfg.sh r06_cfg5 --config dfa256 --batch 131072 --len 4095 --rows 4096 --steps 40 --warmup 8 --distinct 65536 --planes
