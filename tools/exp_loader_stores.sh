#!/bin/bash
# experiment (round 4): cfg 5 with the record stores issued by the LOADER wave instead of the walker (libhrx_exp.so: -DHRX_EXP_LOADER_STORES, wrong output,
# timing only) against the release build, alternating
cd "$(dirname "$0")/.." || exit 1
B="python3 bench.py --config dfa256 --len 4095 --rows 4096 --batch 65536 --steps 20 --warmup 3 --no-verify --no-cpu-baseline --no-pmc --no-spread --allow-debug-flags"
run() { echo -n "$1: "; env $2 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"; }
for i in 1 2 3; do
run "release (walker stores records)" "HRX_X=0"
run "loader stores records          " "HRX_LIB_PATH=$PWD/halo2_regex_amd/csrc/libhrx_exp.so"
done
