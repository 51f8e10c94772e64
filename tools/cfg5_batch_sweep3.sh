#!/bin/bash
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]
print("ms/step %.4f frac %.3f GB %.1f  %s" % (d["ms_per_step"], r["frac"], r["algorithmic_bytes_per_launch"]/1e9, r["launch"]))'
B="python3 bench.py --warmup 2 --no-cpu-baseline --no-pmc --no-verify --no-spread --config dfa256"
for bs in "131072 8191 8192 2 6" "196608 8191 8192 2 4" "393216 2047 2048 2 6" "786432 2047 2048 2 4" "1048576 1023 1024 2 6" "2097152 1023 1024 2 4"; do set -- $bs
echo -n "dfa256 batch $1 x $3: "; timeout 300 $B --len $2 --rows $3 --batch $1 --sets $4 --steps $5 2>/dev/null | python3 -c "$P"; done
