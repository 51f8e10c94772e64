#!/bin/bash
# burst (5 ms idle between graphs of 40 launches) against back-to-back replays, per ablation of the bench-line kernel (libhrx_ablation.so)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export HRX_LIB_PATH=$R/halo2_regex_amd/csrc/libhrx_ablation.so
for f in 0 1 2 3 4 7 0x60 0x800000; do
  for gap in 0 5; do
    echo -n "flags $f gap $gap ms: "; HRX_DEBUG_FLAGS=$f python3 tools/sustained.py 24 40 $gap 2>&1 | grep -v amdgpu.ids | sed 's/.*per replay: //' | awk '{n=NF; s=0; for(i=n-11;i<=n;i++) s+=$i; printf "first %s %s %s ... mean of last 12: %.1f us\n", $1,$2,$3, s/12}'
  done
done
