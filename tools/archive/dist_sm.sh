#!/bin/bash
# String-major shapes in fresh processes (one line per process): the output pair now comes from hrx_alloc_output_pair.
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r02/dist_sm.txt; mkdir -p gpurun_out/r02; : > $O
B="python bench.py --no-cpu-baseline --no-pmc --no-verify --no-spread --layout string-major"
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %8.3f ms frac %.3f %s' % (d['ms_per_step'], r['frac'], r['kernel']))" >> $O; }
for i in 1 2 3; do $B --config regex23 --batch 262144 --len 2047 --rows 2048 --steps 20 --warmup 3 | p cfg3s_sm; done
for i in 1 2; do $B --config dfa256 --len 4095 --rows 4096 --steps 20 --warmup 3 | p cfg5_sm; done
for i in 1 2; do $B --batch 262144 --steps 50 | p regex1x4_sm; done
cat $O
