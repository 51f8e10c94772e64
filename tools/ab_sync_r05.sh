#!/bin/bash
# Ring counters without blocking LDS round trips (ring_post_lds / ring_wait_seen, hrx_device.h) against the build before (libhrx_prev.so, kept beside libhrx.so for the run), same lease, alternating
one() { python bench.py "$@" --steps ${STEPS:-20} --warmup 5 --no-other-configs --no-cpu-baseline --no-pmc --no-spread --allow-debug-flags 2>&1 | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('%.4f ms  frac %.3f' % (l['ms_per_step'], l['roofline']['frac']))"; }
for r in 1 2 3; do
for lib in libhrx_prev.so libhrx.so; do
export HRX_LIB_PATH=$PWD/halo2_regex_amd/csrc/$lib
echo "== $lib"


echo -n "dfa256 65536 x 4096 string-major: "; one --config dfa256 --batch 65536 --rows 4096 --len 4095 --layout string-major
done; done
