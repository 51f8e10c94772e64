#!/bin/bash
B="python3 bench.py --config dfa256 --len 4095 --rows 4096 --warmup 2 --no-cpu-baseline --no-pmc --no-spread"
for bs in "65536 8 20" "131072 8 10" "196608 4 8" "262144 4 6" "524288 2 4" "1048576 2 3"; do set -- $bs
echo -n "batch $1 sets $2: "; timeout 300 $B --batch $1 --sets $2 --steps $3 $( [ $1 -gt 131072 ] && echo --no-verify ) 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step %.4f frac %.3f probe %s verified %s' % (d['ms_per_step'], r['frac'], r.get('no_compute_ms'), (d.get('verified') or {}).get('bit_exact')))"
done
