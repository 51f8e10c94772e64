#!/bin/bash
# Same-process interleaved A/B of the pair-step kernel (forced: 0x40000000) against the one-byte kernel (0x8000000) and
# the planner's choice (0) over batch sizes -> gpurun_out/r02_ab_pair_vs_single.txt (copied to profiles/)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for shape in "16384 1023 1024" "32768 1023 1024" "49152 1023 1024" "65536 1023 1024" "131072 1023 1024" "8192 32768 32769" "65536 4095 4096"; do
  set -- $shape
  HRX_AB_BATCH=$1 HRX_AB_LEN=$2 HRX_AB_ROWS=$3 timeout 300 python3 tools/ab_flags.py 0 0x8000000 0x40000000 2>&1 | grep -v amdgpu.ids
done
