#!/bin/bash
# cfg 5 at 65536 x 4096 (eight direct placement walks, one per buffer set) in fresh processes: the walks' reports next to the rate
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; p=r.get("placement") or {}
print("ms/step %.4f frac %.3f | steps %s best %s ms %s" % (d["ms_per_step"], r["frac"], p.get("steps"), p.get("best_gbs"), [int(x) for x in p.get("search_ms", [])]))'
for i in $(seq ${1:-4}); do timeout 300 python3 bench.py --config dfa256 --len 4095 --rows 4096 --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-spread --no-verify 2>/dev/null | python3 -c "$P"; done
