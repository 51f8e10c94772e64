#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
run() { echo -n "$1: "; shift; python3 bench.py --batch 8192 --len 32768 --rows 32768 --no-cpu-baseline --no-pmc "$@" 2>&1 | python3 -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); print('ms/step %.4f frac %.3f spread %s' % (d['ms_per_step'], d['roofline']['frac'], {k: round(v,4) for k,v in (d.get('spread') or {}).items() if k.startswith('ms')}), d.get('verified',{}).get('bit_exact'))
except Exception as e: print('FAILED', t[-3:])"; }
run "graph sets=2 steps=10" --sets 2 --steps 10 --warmup 3
run "eager sets=2 steps=10" --sets 2 --steps 10 --warmup 3 --eager
run "graph sets=1 steps=20" --sets 1 --steps 20 --warmup 3
HRX_DEBUG_FLAGS=0x80000000 run "sequential (pair-step) sets=2" --sets 2 --steps 10 --warmup 3 --allow-debug-flags
