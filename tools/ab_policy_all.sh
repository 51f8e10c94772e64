#!/bin/bash
# tools/ab_policy.py over the shapes whose finisher / combiner stores masked rows (ntenv build)
cd "$(dirname "$0")/.." || exit 1
export HRX_LIB_PATH=halo2_regex_amd/csrc/libhrx_ntenv.so    # (make -C halo2_regex_amd/csrc ntenv: the release kernels, the store-policy variables read per launch)
T="timeout 400 python3 tools/ab_policy.py"
$T --steps 100 2>/dev/null
$T --dist noise --steps 100 2>/dev/null
$T --config regex23 --batch 262144 --len 2047 --rows 2048 --steps 20 2>/dev/null
$T --config headers3 --batch 65536 --len 2047 --rows 2048 --steps 20 2>/dev/null
$T --config headers3 --batch 32768 --len 32767 --rows 32768 --steps 4 2>/dev/null
$T --config regex123 --steps 50 2>/dev/null
$T --config dfa256 --batch 131072 --len 4095 --rows 4096 --steps 10 2>/dev/null
$T --config dfa256 --batch 131072 --len 4095 --rows 4096 --steps 10 --substr-pairs 0 2>/dev/null
$T --config dfa256 --batch 131072 --len 4095 --rows 4096 --steps 10 --substr-pairs 20 2>/dev/null
$T --config dfa256 --batch 131072 --len 4095 --rows 4096 --steps 10 --substr-pairs 2000 2>/dev/null
