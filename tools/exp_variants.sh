#!/bin/bash
# round 4: cfg 5 under release-based experiment builds (libhrx_exp_<tag>.so: one compile-time macro each, wrong output, timing only), alternating with the release build
cd "$(dirname "$0")/.." || exit 1
B="python3 bench.py --config dfa256 --len 4095 --rows 4096 --batch ${BATCH:-65536} --steps 20 --warmup 3 --no-verify --no-cpu-baseline --no-pmc --allow-debug-flags"
run() { echo -n "$1: "; env $2 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; mc=r.get('mix_ceiling') or {}
print('ms/step %.4f frac %.3f  spread median %.4f  traffic pass %.4f' % (d['ms_per_step'], r['frac'], (d.get('spread') or {}).get('ms_per_step_median', 0), (mc.get('traffic_pass_us') or 0) / 1e3))"; }
for i in 1 2; do
run "release                      " "HRX_X=0"
for t in "$@"; do run "$(printf '%-29s' $t)" "HRX_LIB_PATH=$PWD/halo2_regex_amd/csrc/libhrx_exp_$t.so"; done
done
