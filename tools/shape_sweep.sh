#!/bin/bash
# planner sanity over many batch shapes (position-major, 2 rotating sets, not verified): fraction of the 8 TB/s peak per (config, strings, rows)
cd "$(dirname "$0")/.." || exit 1
for cfg in regex1 regex23 headers3 dfa256; do for m in 1024 4096; do
echo -n "$cfg M=$m:"
for b in 4096 12288 20480 40000 65536 100000 200000; do
python3 bench.py --config $cfg --batch $b --len $((m-1)) --rows $m --sets 2 --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-spread --no-verify 2>/dev/null | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; k=r['kernel']
    tag='pp' if 'pp_kernel' in k else 'pmd' if 'pmd' in k else 'byte' if 'true>' in k and 'false, false, false, false' in k else 'pm'
    print(' %d:%.2f(%s%s)' % ($b, r['frac'], tag, '+c' if 'chunked' in r.get('launch','')+k else ''), end='')
except Exception as e: print(' $b:FAIL', end='')"
done; echo; done; done
