#!/bin/bash
# A/B of library builds on cfg 5 (profiling only): tools/ab_cfg5.sh lib1.so lib2.so ...
cd "$(dirname "$0")/.." || exit 1
for i in 1 2; do for v in "$@"; do for shape in "65536" "131072"; do
  echo -n "$v B=$shape: "; HRX_LIB_PATH=$PWD/halo2_regex_amd/csrc/$v python3 bench.py --config dfa256 --len 4096 --rows 4096 --batch $shape --sets 1 --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-spread --allow-debug-flags 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step %.4f frac %.3f verified %s' % (d['ms_per_step'], d['roofline']['frac'], d['verified']['bit_exact']))"
done; done; done
