#!/bin/bash
# alternate two builds of the library over fresh processes (the launch time has a per-process state): tools/ab_libs.sh <libA> <libB> <reps> <bench args...>
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
A=$1; Bl=$2; reps=$3; shift 3
for i in $(seq $reps); do for lib in $A $Bl; do
  echo -n "$lib: "; HRX_LIB_PATH=$R/halo2_regex_amd/csrc/$lib python3 bench.py --no-cpu-baseline --no-verify --no-spread "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.4f ms  frac %.3f  %s' % (d['ms_per_step'], r['frac'], r['kernel']))"
done; done
