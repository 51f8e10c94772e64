#!/bin/bash
# quick look after a kernel change: parity, then the bench-line kernel back to back / in bursts, release and ablation builds
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3)
for gap in 0 5; do echo -n "release gap $gap: "; python3 tools/sustained.py 24 40 $gap 2>&1 | grep -v amdgpu.ids | sed 's/.*per replay: //' | awk '{n=NF; s=0; for(i=n-11;i<=n;i++) s+=$i; printf "first %s %s %s ... mean of last 12: %.1f us\n", $1,$2,$3, s/12}'; done
export HRX_LIB_PATH=$R/halo2_regex_amd/csrc/libhrx_ablation.so
for f in ${QUICK_FLAGS:-0 3 7}; do for gap in 0 5; do
  echo -n "ablation flags $f gap $gap: "; HRX_DEBUG_FLAGS=$f python3 tools/sustained.py 24 40 $gap 2>&1 | grep -v amdgpu.ids | sed 's/.*per replay: //' | awk '{n=NF; s=0; for(i=n-11;i<=n;i++) s+=$i; printf "first %s %s %s ... mean of last 12: %.1f us\n", $1,$2,$3, s/12}'
done; done
