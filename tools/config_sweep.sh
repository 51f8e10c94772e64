#!/bin/bash
# One bench line per BASELINE config shape (per-GPU sizes) -> gpurun_out/r06_config_sweep/*.json; copied to profiles/r06_config_sweep/.
# Round 6: every position-major line of two or more defs twice — interleaved records and RECORD PLANES (--planes: buffers from hrx_alloc_output_planes_for_batch); cfg 5 also
# on its one records buffer chosen the same way.
# Every line in the HBM-only regime (the timed steps rotate over as many buffer sets as fit 48 GiB, up to 8) and verified over every string.
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r06_config_sweep${SWEEP_TAG:-}; rm -rf $O; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-pmc"
$B                                                                                   > $O/cfg2_regex1_65536x1024.json
$B --layout string-major                                                             > $O/cfg2_regex1_65536x1024_string_major.json
$B --batch 131072                                                                    > $O/regex1_131072x1024.json
$B --config regex23 --batch 262144 --len 2047 --rows 2048 --steps 20 --warmup 3      > $O/cfg3_regex23_262144x2048.json
$B --config regex23 --batch 1048576 --len 2047 --rows 2048 --steps 5 --warmup 2 --distinct 65536 > $O/cfg3_regex23_1048576x2048_full.json
$B --config regex23 --batch 262144 --len 2047 --rows 2048 --steps 20 --warmup 3 --layout string-major > $O/cfg3_regex23_262144x2048_string_major.json
$B --config headers3 --batch 32768 --len 32767 --rows 32768 --steps 5 --warmup 2 --distinct 4096 > $O/cfg4_headers3_32768x32768.json
$B --config headers3 --batch 65536 --len 2047 --rows 2048 --steps 20 --warmup 3      > $O/headers3_65536x2048.json
$B --config headers5 --batch 65536 --len 2047 --rows 2048 --steps 20 --warmup 3      > $O/headers5_65536x2048.json
$B --config headers5 --batch 65536 --len 1023 --rows 1024 --steps 20 --warmup 3 --layout string-major > $O/headers5_65536x1024_string_major.json
$B --config headers5 --batch 65536 --len 1023 --rows 1024 --steps 20 --warmup 3 --layout string-major --dense > $O/headers5_65536x1024_string_major_dense_pitches.json
$B --config headers5 --batch 65536 --len 2047 --rows 2048 --steps 20 --warmup 3 --layout string-major > $O/headers5_65536x2048_string_major.json
$B --config headers4 --batch 65536 --len 1023 --rows 1024 --steps 20 --warmup 3 --layout string-major > $O/headers4_65536x1024_string_major.json
$B --config headers4 --batch 65536 --len 2047 --rows 2048 --steps 20 --warmup 3                      > $O/headers4_65536x2048.json
$B --config regex123 --steps 50                                                      > $O/regex123_65536x1024.json
$B --config dfa256 --len 4095 --rows 4096 --steps 20 --warmup 3                      > $O/cfg5_dfa256_65536x4096.json
$B --config dfa256 --batch 131072 --len 4095 --rows 4096 --steps 40 --warmup 8 --distinct 65536 > $O/cfg5_dfa256_131072x4096.json
$B --config dfa256 --batch 131072 --len 4095 --rows 4096 --steps 40 --warmup 8 --distinct 65536 --planes > $O/cfg5_dfa256_131072x4096_chosen_with_the_batch.json
$B --config dfa256 --batch 131072 --len 4095 --rows 4096 --steps 40 --warmup 8 --distinct 65536 --substr-defs 2 --substr-pairs 100 > $O/cfg5_dfa256_two_substr_defs_131072x4096.json
$B --config dfa256 --len 4095 --rows 4096 --steps 20 --warmup 3 --layout string-major > $O/cfg5_dfa256_65536x4096_string_major.json
$B --batch 8192 --len 32767 --rows 32768 --steps 10 --warmup 2 --distinct 2048                       > $O/regex1_8192x32768_long.json
$B --batch 16384 --len 32767 --rows 32768 --steps 10 --warmup 2 --distinct 2048                      > $O/regex1_16384x32768_long.json
$B --config headers3 --batch 8192 --len 32767 --rows 32768 --steps 5 --warmup 2 --distinct 2048      > $O/headers3_8192x32768_long.json
$B --config regex23 --batch 8192 --len 32767 --rows 32768 --steps 5 --warmup 2 --distinct 2048       > $O/regex23_8192x32768_long.json
$B --config regex23 --batch 262144 --len 2047 --rows 2048 --steps 20 --warmup 3 --planes      > $O/cfg3_regex23_262144x2048_planes.json
$B --config regex23 --batch 1048576 --len 2047 --rows 2048 --steps 5 --warmup 2 --distinct 65536 --planes > $O/cfg3_regex23_1048576x2048_full_planes.json
$B --config headers3 --batch 32768 --len 32767 --rows 32768 --steps 5 --warmup 2 --distinct 4096 --planes > $O/cfg4_headers3_32768x32768_planes.json
$B --config headers3 --batch 65536 --len 2047 --rows 2048 --steps 20 --warmup 3 --planes      > $O/headers3_65536x2048_planes.json
$B --config headers5 --batch 65536 --len 2047 --rows 2048 --steps 20 --warmup 3 --planes      > $O/headers5_65536x2048_planes.json
$B --config headers4 --batch 65536 --len 2047 --rows 2048 --steps 20 --warmup 3 --planes      > $O/headers4_65536x2048_planes.json
$B --config regex123 --steps 50 --planes                                                      > $O/regex123_65536x1024_planes.json
$B --config headers3 --batch 8192 --len 32767 --rows 32768 --steps 5 --warmup 2 --distinct 2048 --planes > $O/headers3_8192x32768_long_planes.json
$B --config regex23 --batch 8192 --len 32767 --rows 32768 --steps 5 --warmup 2 --distinct 2048 --planes  > $O/regex23_8192x32768_long_planes.json
HRX_DEBUG_FLAGS=0x80000000 $B --allow-debug-flags --batch 8192 --len 32767 --rows 32768 --steps 10 --warmup 2 --distinct 2048 > $O/regex1_8192x32768_long_sequential.json
HRX_DEBUG_FLAGS=0x80000000 $B --allow-debug-flags --config headers3 --batch 8192 --len 32767 --rows 32768 --steps 5 --warmup 2 --distinct 2048 > $O/headers3_8192x32768_long_sequential.json
for f in $O/*.json; do python3 -c "
import sys,json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; mc=r.get('mix_ceiling') or {}
    print('%-50s %8.3f ms  %.3e rows/s  frac %.3f  probe %s  sets %s  verified %s  %s' % ('$(basename $f .json)', d['ms_per_step'], d['value'], r['frac'], '%.3f ms' % (mc['traffic_pass_us'] / 1e3) if mc.get('traffic_pass_us') else '-', d['config']['buffer_sets'][:2], (d.get('verified') or {}).get('bit_exact'), r['kernel'][:70]))
except Exception as e: print('$(basename $f .json)', 'FAILED', e)"; done | tee $O/summary.txt
