#!/bin/bash
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; p=r.get("placement") or {}
print("ms/step %.4f frac %.3f GB %.1f  %s | placement steps %s chosen %s ref %s first %s best %s" % (d["ms_per_step"], r["frac"], r["algorithmic_bytes_per_launch"]/1e9, r["launch"], p.get("steps"), p.get("chosen_step"), p.get("ref_gbs"), p.get("first_gbs"), p.get("best_gbs")))'
B="python3 bench.py --warmup 2 --no-cpu-baseline --no-pmc --no-verify --no-spread --config dfa256"
for bs in "262144 4095 4096 2 6" "327680 4095 4096 2 6" "393216 4095 4096 2 4" "393216 4095 4096 1 4"; do set -- $bs
echo -n "dfa256 batch $1 x $3 sets $4: "; timeout 300 $B --len $2 --rows $3 --batch $1 --sets $4 --steps $5 2>/dev/null | python3 -c "$P"; done
echo -n "plain allocations (HRX_PLACE=0) 393216: "; HRX_PLACE=0 timeout 300 $B --len 4095 --rows 4096 --batch 393216 --sets 1 --steps 4 2>/dev/null | python3 -c "$P"
