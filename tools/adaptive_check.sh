#!/bin/bash
# the per-tile streaming / write-back choice of the masked rows (hrx_kernel_pm.hip octets_out): cfg 5 against the number of tagged pairs, and the shapes whose finisher shares the code
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]
print("ms/step %.4f frac %.3f verified %s" % (d["ms_per_step"], r["frac"], (d.get("verified") or {}).get("bit_exact")))'
B="python3 bench.py --warmup 3 --no-cpu-baseline --no-pmc --no-spread"
for p in 0 20 200 2000; do echo -n "cfg5 131072 pairs $p: "; timeout 300 $B --config dfa256 --len 4095 --rows 4096 --batch 131072 --steps 10 --substr-pairs $p 2>/dev/null | python3 -c "$P"; done
echo -n "cfg5 65536 pairs 200: "; timeout 300 $B --config dfa256 --len 4095 --rows 4096 --steps 20 2>/dev/null | python3 -c "$P"
echo -n "cfg2 headline: "; timeout 300 $B --warmup 20 2>/dev/null | python3 -c "$P"
echo -n "cfg3 262144x2048: "; timeout 300 $B --config regex23 --batch 262144 --len 2047 --rows 2048 --steps 20 2>/dev/null | python3 -c "$P"
echo -n "headers3 65536x2048: "; timeout 300 $B --config headers3 --batch 65536 --len 2047 --rows 2048 --steps 20 2>/dev/null | python3 -c "$P"
echo -n "cfg2 noise dist: "; timeout 300 $B --dist noise --warmup 20 2>/dev/null | python3 -c "$P"
