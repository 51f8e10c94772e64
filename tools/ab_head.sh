#!/bin/bash
# A/B of two library builds on one box (profiling only): stamps of the rotating regime + the bench line, alternating
cd "$(dirname "$0")/.." || exit 1
L=halo2_regex_amd/csrc
for i in 1 2; do
  for v in ab/libhrx_stamps_old.so libhrx_stamps.so; do
    echo "== stamps $v"; HRX_PLACE=0 HRX_LIB_PATH=$L/$v python3 tools/stamps_rotating.py 8 40 | grep -E "launches|start walking|walker done"
  done
done
for i in 1 2 3; do
  for v in ab/libhrx_old.so libhrx.so; do
    echo "== bench $v"; HRX_LIB_PATH=$PWD/$L/$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --allow-debug-flags 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step %.4f frac %.3f spread-median %.4f probe %.1f one-set %.4f' % (d['ms_per_step'], r['frac'], d['spread']['ms_per_step_median'], r['mix_ceiling']['traffic_pass_us'], r['one_buffer_set']['ms_per_step_median']))"
  done
done
