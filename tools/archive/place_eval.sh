#!/bin/bash
# Placement strategies side by side, fresh processes, one box: plain allocations | placed pair | two-class chunks | what the library
# picks by measurement.  Needs tools/experimental/two_class_chunks.patch applied (HRX_PLACE is honoured by that build's
# libhrx_ablation.so only).
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r02/place_eval_$1.txt; mkdir -p gpurun_out/r02; : > $O
export HRX_LIB_PATH=$PWD/halo2_regex_amd/csrc/libhrx_ablation.so
B="python bench.py --no-cpu-baseline --no-pmc --no-verify --no-spread"
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%-10s %-8s %8.3f ms frac %.3f' % ('$1', '$2', d['ms_per_step'], r['frac']))" >> $O; }
for rep in 1 2; do
for m in plain pair chunks auto; do
  if [ $m = auto ]; then unset HRX_PLACE; export HRX_PLACE_TRACE=1; else export HRX_PLACE=$m; unset HRX_PLACE_TRACE; fi
  $B --config regex23 --batch 262144 --len 2047 --rows 2048 --steps 20 --warmup 3 2>> $O | p cfg3s $m
  $B --config headers3 --batch 65536 --len 2047 --rows 2048 --steps 20 --warmup 3 2>> $O | p h3s $m
  $B --config headers3 --batch 32768 --len 32767 --rows 32768 --steps 5 --warmup 2 2>> $O | p cfg4 $m
  $B --config dfa256 --batch 131072 --len 4095 --rows 4096 --steps 10 --warmup 3 2>> $O | p cfg5x2 $m
done; done
grep -v "masked-row candidate" $O | sort -k1,1 -s
