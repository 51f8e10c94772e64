#!/bin/bash
# cfg 2 (regex1, 65536 x 1024 B), string-major: ablations of the walker/storer kernel (profiling only; `make ablation`)
cd "$(dirname "$0")/.." || exit 1
L=$PWD/halo2_regex_amd/csrc/libhrx_ablation.so
B="python3 bench.py --layout string-major --sets 8 --steps 100 --warmup 3 --no-verify --no-cpu-baseline --no-pmc --no-spread --allow-debug-flags"
run() { echo -n "$1: "; env HRX_LIB_PATH=$L $2 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"; }
for i in 1 2; do
run "shipped                         " "HRX_DEBUG_FLAGS=0"
run "masked stores skipped           " "HRX_DEBUG_FLAGS=2"
run "records skipped                 " "HRX_DEBUG_FLAGS=1"
run "both skipped                    " "HRX_DEBUG_FLAGS=3"
run "no walk (storer only)           " "HRX_DEBUG_FLAGS=16"
run "no touch-ahead                  " "HRX_DEBUG_FLAGS=8"
run "write-back stores               " "HRX_DEBUG_FLAGS=96"
run "input from L2                   " "HRX_DEBUG_FLAGS=4"
done
