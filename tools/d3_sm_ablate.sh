#!/bin/bash
# D = 3 string-major (hrx::witness_pm_kernel<3, false, true, false, true>: lane-direct 16-byte stores), regex123 65536 x 1024 B: ablations (`make ablation`)
cd "$(dirname "$0")/.." || exit 1
L=$PWD/halo2_regex_amd/csrc/libhrx_ablation.so
B="python3 bench.py --config regex123 --layout string-major --sets 4 --steps 30 --warmup 3 --no-verify --no-cpu-baseline --no-pmc --no-spread --allow-debug-flags"
run() { echo -n "$1: "; env HRX_LIB_PATH=$L $2 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"; }
run "shipped                       " "HRX_DEBUG_FLAGS=0"
run "records skipped               " "HRX_DEBUG_FLAGS=1"
run "masked stores skipped         " "HRX_DEBUG_FLAGS=2"
run "both skipped                  " "HRX_DEBUG_FLAGS=3"
run "stores onto the first lines   " "HRX_DEBUG_FLAGS=0x1000000"
B="${B/--layout string-major/}"; run "position-major (for reference)" "HRX_DEBUG_FLAGS=0"
