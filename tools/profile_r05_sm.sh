#!/bin/bash
# Round 5: rocprofv3 kernel stats + PMC traffic for the string-major rows out of the def-parallel launch (hrx_kernel_pmd.hip SMO), five and four defs, 65536 x 1024 B, recommended pitches
# -> gpurun_out/r05_d5sm_{kernel_stats.csv,pmc.json}, r05_d4sm_* (copied to profiles/)
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/profile_cfg.sh r05_d5sm --config headers5 --layout string-major --steps 20 --warmup 5
bash tools/profile_cfg.sh r05_d4sm --config headers4 --layout string-major --steps 20 --warmup 5
