#!/bin/bash
# The shapes whose fraction depends on the box (cfg 3 full, cfg 4 share, cfg 5), one lease each: tools/box_sweep.sh <tag> -> gpurun_out/r05_box_<tag>/*.json + summary
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r05_box_$1; rm -rf $O; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-pmc"
$B --steps 20 --warmup 5                                                             > $O/cfg2_regex1_65536x1024.json
$B --config regex23 --batch 1048576 --len 2047 --rows 2048 --steps 5 --warmup 2 --distinct 65536      > $O/cfg3_regex23_1048576x2048_full.json
$B --config headers3 --batch 32768 --len 32767 --rows 32768 --steps 5 --warmup 2 --distinct 4096     > $O/cfg4_headers3_32768x32768.json
$B --config dfa256 --batch 131072 --len 4095 --rows 4096 --steps 10 --warmup 3 --distinct 65536       > $O/cfg5_dfa256_131072x4096.json
for f in $O/*.json; do python3 -c "
import sys,json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; mc=r.get('mix_ceiling') or {}; pl=r.get('placement') or {}
    print('%-40s %8.3f ms  frac %.3f  spread median %.3f  probe %s  verified %s  placement best GB/s %s steps %s' % ('$(basename $f .json)', d['ms_per_step'], r['frac'], (d.get('spread') or {}).get('ms_per_step_median', 0), '%.3f ms' % (mc['traffic_pass_us'] / 1e3) if mc.get('traffic_pass_us') else '-', (d.get('verified') or {}).get('bit_exact'), pl.get('best_gbs'), pl.get('steps')))
except Exception as e: print('$(basename $f .json)', 'FAILED', e)"; done | tee $O/summary.txt
