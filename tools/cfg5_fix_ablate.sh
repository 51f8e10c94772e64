#!/bin/bash
# cfg 5: what the end-mask repairs cost (ablation build, alternating): shipped / repairs skipped / repairs stored onto hot lines
cd "$(dirname "$0")/.." || exit 1
L=$PWD/halo2_regex_amd/csrc/libhrx_ablation.so
B="python3 bench.py --config dfa256 --len 4095 --rows 4096 --batch ${1:-65536} --steps 20 --warmup 3 --no-verify --no-cpu-baseline --no-pmc --no-spread --allow-debug-flags"
run() { echo -n "$1: "; env HRX_LIB_PATH=$L $2 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"; }
for i in 1 2 3 4; do
run "shipped                 " "HRX_DEBUG_FLAGS=0"
run "repairs skipped         " "HRX_DEBUG_FLAGS=0x800000"
run "repairs onto hot lines  " "HRX_DEBUG_FLAGS=0x4000"
done
