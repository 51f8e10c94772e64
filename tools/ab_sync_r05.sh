#!/bin/bash
one() { python bench.py "$@" --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-pmc --no-spread --allow-debug-flags 2>&1 | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('%.4f ms  frac %.3f verified %s' % (l['ms_per_step'], l['roofline']['frac'], l['config'].get('bit_exact')))"; }
for r in 1 2; do
for lib in libhrx_prev.so libhrx.so; do
export HRX_LIB_PATH=$PWD/halo2_regex_amd/csrc/$lib
echo "== $lib"
echo -n "headers5 PM: "; one --config headers5
echo -n "headers5 SM: "; one --config headers5 --layout string-major --dense
echo -n "headers3 32768x8192: "; one --config headers3 --batch 32768 --rows 8192 --len 8191 --distinct 4096
done; done
