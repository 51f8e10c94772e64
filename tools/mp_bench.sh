#!/bin/bash
# more than three defs (headers5, D = 5): the last pass merging the groups' summaries itself against the separate combine launch
cd "$(dirname "$0")/.." || exit 1
run() { echo -n "$1: "; shift; env $ENVV python3 bench.py --config headers5 --batch 65536 --no-cpu-baseline --no-pmc --no-spread "$@" 2>&1 | python3 -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); r=d['roofline']; print('ms/step %.4f frac %.3f verified %s  %s' % (d['ms_per_step'], r['frac'], (d.get('verified') or {}).get('bit_exact'), r['kernel'][-90:]))
except Exception as e: print('FAILED', t[-3:])"; }
for i in 1 2; do
ENVV="HRX_MP_COMBINE=0" run "merge in the last pass, pm 2048" --len 2047 --rows 2048 --steps 20 --warmup 3
ENVV="HRX_MP_COMBINE=1" run "combine launch,          pm 2048" --len 2047 --rows 2048 --steps 20 --warmup 3
ENVV="HRX_MP_COMBINE=0" run "merge in the last pass, sm 1024" --len 1023 --rows 1024 --steps 20 --warmup 3 --layout string-major
ENVV="HRX_MP_COMBINE=1" run "combine launch,          sm 1024" --len 1023 --rows 1024 --steps 20 --warmup 3 --layout string-major
done
