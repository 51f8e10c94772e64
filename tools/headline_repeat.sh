#!/bin/bash
# the bench line N times in fresh processes on one lease: the arena walk's report next to the rate (does a walk ever settle for a colliding pairing?)
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; p=r.get("placement") or {}
print("ms/step %.4f frac %.3f | steps %s chosen %s ref %s first %s best %s accepted %s search_ms %s" % (d["ms_per_step"], r["frac"], p.get("steps",[None])[0], p.get("chosen_step",[None])[0], p.get("ref_gbs",[None])[0], p.get("first_gbs",[None])[0], p.get("best_gbs",[None])[0], p.get("sets_accepted"), p.get("search_ms",[None])[0]))'
for i in $(seq ${1:-8}); do timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-spread --no-verify 2>/dev/null | python3 -c "$P"; done
