#!/bin/bash
# cfg 5 (256-state DFA), string-major: ablations of the walker/storer kernel on the BYTE table (profiling only; `make ablation`)
cd "$(dirname "$0")/.." || exit 1
L=$PWD/halo2_regex_amd/csrc/libhrx_ablation.so
B="python3 bench.py --config dfa256 --len 4095 --rows 4096 --batch 65536 --layout string-major --sets 2 --steps 20 --warmup 3 --no-verify --no-cpu-baseline --no-pmc --no-spread --allow-debug-flags"
run() { echo -n "$1: "; env HRX_LIB_PATH=$L $2 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"; }
for i in 1 2; do
run "shipped                         " "HRX_DEBUG_FLAGS=0"
run "fix-ups skipped                 " "HRX_DEBUG_FLAGS=0x800000"
run "masked stores skipped           " "HRX_DEBUG_FLAGS=2"
run "fix-ups + masked stores skipped " "HRX_DEBUG_FLAGS=0x800002"
run "records skipped                 " "HRX_DEBUG_FLAGS=1"
run "all three skipped               " "HRX_DEBUG_FLAGS=0x800003"
run "no walk (storer only)           " "HRX_DEBUG_FLAGS=16"
run "no touch-ahead                  " "HRX_DEBUG_FLAGS=8"
done
