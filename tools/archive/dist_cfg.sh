#!/bin/bash
# Launch-time distribution over fresh processes for the cfg 3 / cfg 4 shapes (one line per process).
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r02/dist_cfg.txt; mkdir -p gpurun_out/r02; : > $O
B="python bench.py --no-cpu-baseline --no-pmc --no-verify --no-spread"
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %8.3f ms frac %.3f fresh %s %s' % (d['ms_per_step'], r['frac'], (r.get('fresh_buffers') or {}).get('frac'), r['kernel']))" >> $O; }
for i in 1 2 3; do $B --config headers3 --batch 32768 --len 32767 --rows 32768 --steps 5 --warmup 2 | p cfg4; done
for i in 1 2 3; do $B --config regex23 --batch 262144 --len 2047 --rows 2048 --steps 20 --warmup 3 | p cfg3s; done
for i in 1 2 3; do $B --config headers3 --batch 65536 --len 2047 --rows 2048 --steps 20 --warmup 3 | p h3s; done
for i in 1 2 3; do $B --config regex23 --batch 1048576 --len 2047 --rows 2048 --steps 5 --warmup 2 | p cfg3full; done
for i in 1 2; do $B --config dfa256 --batch 131072 --len 4095 --rows 4096 --steps 10 --warmup 3 | p cfg5x2; done
for i in 1 2; do $B --batch 262144 --steps 50 | p regex1x4; done
cat $O
