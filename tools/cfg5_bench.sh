#!/bin/bash
# cfg 5 (dfa256) bench lines: position-major and string-major, 65536 and 131072 strings
cd "$(dirname "$0")/.." || exit 1
run() { echo -n "$1: "; shift; python3 bench.py --config dfa256 --len 4095 --rows 4096 --no-cpu-baseline --no-pmc --no-spread "$@" 2>&1 | python3 -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); r=d['roofline']; print('ms/step %.4f frac %.3f one_set %s  %s  verified %s' % (d['ms_per_step'], r['frac'], round((r.get('one_buffer_set') or {}).get('frac') or 0, 3), r['kernel'][:64], (d.get('verified') or {}).get('bit_exact')), (r.get('placement') or {}).get('best_gbs'))
except Exception as e: print('FAILED', t[-3:])"; }
run "pm 65536" --steps 20 --warmup 3
run "pm 131072" --batch 131072 --steps 10 --warmup 3
run "sm 65536" --steps 20 --warmup 3 --layout string-major
run "sm 131072" --batch 131072 --steps 10 --warmup 3 --layout string-major
