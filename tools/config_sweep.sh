#!/bin/bash
# One bench line per BASELINE config shape (per-GPU sizes) -> gpurun_out/r02_config_sweep/*.json; copied to profiles/r02_config_sweep/.
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r02_config_sweep; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-pmc"
$B                                                                                   > $O/cfg2_regex1_65536x1024.json
$B --layout string-major                                                             > $O/cfg2_regex1_65536x1024_string_major.json
$B --batch 131072                                                                    > $O/regex1_131072x1024.json
$B --config regex23 --batch 262144 --len 2047 --rows 2048 --steps 20 --warmup 3      > $O/cfg3_regex23_262144x2048.json
$B --config regex23 --batch 1048576 --len 2047 --rows 2048 --steps 5 --warmup 2      > $O/cfg3_regex23_1048576x2048_full.json
$B --config regex23 --batch 262144 --len 2047 --rows 2048 --steps 20 --warmup 3 --layout string-major > $O/cfg3_regex23_262144x2048_string_major.json
$B --config headers3 --batch 32768 --len 32767 --rows 32768 --steps 5 --warmup 2     > $O/cfg4_headers3_32768x32768.json
$B --config headers3 --batch 65536 --len 2047 --rows 2048 --steps 20 --warmup 3      > $O/headers3_65536x2048.json
$B --config regex123 --steps 50                                                      > $O/regex123_65536x1024.json
$B --config dfa256 --len 4095 --rows 4096 --steps 20 --warmup 3                      > $O/cfg5_dfa256_65536x4096.json
$B --config dfa256 --batch 131072 --len 4095 --rows 4096 --steps 10 --warmup 3       > $O/cfg5_dfa256_131072x4096.json
$B --config dfa256 --len 4095 --rows 4096 --steps 20 --warmup 3 --layout string-major > $O/cfg5_dfa256_65536x4096_string_major.json
$B --batch 8192 --len 32767 --rows 32768 --steps 10 --warmup 2                       > $O/regex1_8192x32768_long.json
for f in $O/*.json; do python -c "import sys,json; d=json.loads(open('$f').read()); r=d['roofline']; print('%-48s %8.3f ms  %.3e rows/s  frac %.3f  copy-ceil %.0f  %s' % ('$(basename $f .json)', d['ms_per_step'], d['value'], r['frac'], 0, r['kernel']))"; done
