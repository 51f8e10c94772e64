#!/bin/bash
# cfg 5 position-major, two library builds alternating on ONE lease (leases differ by +-3 %): tools/ab_cfg5_libs.sh <other .so> [rounds]
cd "$(dirname "$0")/.." || exit 1
OTHER=$1; N=${2:-3}
B="python3 bench.py --config dfa256 --len 4095 --rows 4096 --warmup 3 --no-cpu-baseline --no-pmc --no-spread"
run() { echo -n "$1: "; env $2 $B $3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step %.4f frac %.3f verified %s' % (d['ms_per_step'], r['frac'], (d.get('verified') or {}).get('bit_exact')))"; }
for i in $(seq $N); do
run "release 65536 " "HRX_X=0" "--steps 20"
run "other   65536 " "HRX_LIB_PATH=$OTHER" "--steps 20 --allow-debug-flags"
run "release 131072" "HRX_X=0" "--batch 131072 --steps 10"
run "other   131072" "HRX_LIB_PATH=$OTHER" "--batch 131072 --steps 10 --allow-debug-flags"
done
