#!/bin/bash
# two library builds alternating on ONE lease over several shapes: tools/ab_libs.sh <other .so> [rounds]
cd "$(dirname "$0")/.." || exit 1
OTHER=$1; N=${2:-2}
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]
print("ms/step %.4f frac %.3f verified %s" % (d["ms_per_step"], r["frac"], (d.get("verified") or {}).get("bit_exact")))'
B="python3 bench.py --no-cpu-baseline --no-pmc --no-spread"
run() { echo -n "$1 | $2: "; shift; local lib=$1; shift; if [ "$lib" = other ]; then HRX_LIB_PATH=$OTHER timeout 400 $B --allow-debug-flags "$@" 2>/dev/null | python3 -c "$P"; else timeout 400 $B "$@" 2>/dev/null | python3 -c "$P"; fi; }
for i in $(seq $N); do
for lib in release other; do
run "cfg2 regex1 65536x1024    " $lib --warmup 20
run "cfg3 regex23 262144x2048  " $lib --config regex23 --batch 262144 --len 2047 --rows 2048 --steps 20 --warmup 3
run "cfg4 headers3 32768x32768 " $lib --config headers3 --batch 32768 --len 32767 --rows 32768 --steps 5 --warmup 2
run "headers3 65536x2048       " $lib --config headers3 --batch 65536 --len 2047 --rows 2048 --steps 20 --warmup 3
run "regex123 65536x1024       " $lib --config regex123 --steps 50 --warmup 3
run "cfg5 dfa256 131072x4096   " $lib --config dfa256 --batch 131072 --len 4095 --rows 4096 --steps 10 --warmup 3
done; done
