#!/bin/bash
# the streaming / write-back mix of the position-major kernel's stores (HRX_NT_MIX, ablation build) over fresh processes:
# tools/nt_mix_sweep.sh <reps> <mix values...> [-- bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
reps=$1; shift; mixes=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do mixes+=($1); shift; done; [ "$1" == "--" ] && shift
for i in $(seq $reps); do for m in "${mixes[@]}"; do
  echo -n "nt_mix $m: "; HRX_NT_MIX=$m HRX_LIB_PATH=$R/halo2_regex_amd/csrc/libhrx_ablation.so python3 bench.py --no-cpu-baseline --no-verify --no-spread --no-pmc --allow-debug-flags "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.4f ms  frac %.3f' % (d['ms_per_step'], r['frac']))"
done; done | sort -s -k2,2
