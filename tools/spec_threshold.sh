#!/bin/bash
# where the chunked launch stops paying: few long strings, 1 .. 2 groups of 64 strings per CU, chunked (the planner's choice below two / 1.75 / 1.5
# groups per CU at D = 1 / 2 / 3) against the sequential kernels (HRX_DEBUG_FLAGS=0x80000000: never chunked)
cd "$(dirname "$0")/.." || exit 1
run() { python3 bench.py --allow-debug-flags --config $1 --batch $2 --len 32767 --rows 32768 --steps 4 --warmup 2 --sets 3 --no-cpu-baseline --no-pmc --no-spread --no-verify 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%s B=%s %s: ms/step %.4f frac %.3f %s' % ('$1', '$2', '$3', d['ms_per_step'], r['frac'], r['kernel'][:44]))"; }
for cfg in regex1 regex23 headers3; do for b in 20480 24576 28672 32768; do
run $cfg $b "planner   "
HRX_DEBUG_FLAGS=0x80000000 run $cfg $b sequential
done; done
# shorter strings (no chunking below 4096 rows): the pair-step kernel between one and two groups per CU
for b in 20480 24576; do python3 bench.py --batch $b --no-cpu-baseline --no-pmc --no-spread --no-verify 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('regex1 B=$b x 1024: ms/step %.4f frac %.3f %s' % (d['ms_per_step'], r['frac'], r['kernel'][:44]), r['launch'])"; done
