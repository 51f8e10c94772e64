#!/bin/bash
# Ring counters without blocking LDS round trips (ring_post_lds / ring_wait_seen, hrx_device.h) against the build before (libhrx_prev.so, kept beside libhrx.so for the run), same lease, alternating
one() { python bench.py "$@" --steps ${STEPS:-20} --warmup 5 --no-other-configs --no-cpu-baseline --no-pmc --no-spread --allow-debug-flags 2>&1 | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('%.4f ms  frac %.3f' % (l['ms_per_step'], l['roofline']['frac']))"; }
for r in 1 2 3; do
for lib in libhrx_prev.so libhrx.so; do
export HRX_LIB_PATH=$PWD/halo2_regex_amd/csrc/$lib
echo "== $lib"
echo -n "regex1 65536 x 1024 (headline): "; STEPS=200 one
echo -n "regex23 1048576 x 2048: "; one --config regex23 --batch 1048576 --rows 2048 --len 2047 --distinct 65536
echo -n "dfa256 131072 x 4096: "; one --config dfa256 --batch 131072 --rows 4096 --len 4095 --distinct 65536
echo -n "regex1 string-major: "; one --layout string-major
done; done
