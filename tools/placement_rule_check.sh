#!/bin/bash
# the placement walk's acceptance rule on the shapes whose rate depends on it: kept pairing and steps per buffer set, and the rate of the launch
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; p=r.get("placement") or {}
print("ms/step %.4f frac %.3f | steps %s chosen %s ref %s first %s best %s search_ms %s" % (d["ms_per_step"], r["frac"], p.get("steps"), p.get("chosen_step"), p.get("ref_gbs"), p.get("first_gbs"), p.get("best_gbs"), p.get("search_ms")))'
B="python3 bench.py --warmup 2 --no-cpu-baseline --no-pmc --no-verify --no-spread"
run() { echo -n "$1: "; shift; timeout 400 $B "$@" 2>/dev/null | python3 -c "$P"; }
run "cfg2 regex1 65536x1024        " --steps 200 --warmup 20
run "cfg5 dfa256 131072x4096       " --config dfa256 --len 4095 --rows 4096 --batch 131072 --steps 10
run "cfg5 dfa256 393216x4096 sets 2" --config dfa256 --len 4095 --rows 4096 --batch 393216 --sets 2 --steps 4
run "cfg5 dfa256 524288x4096 sets 2" --config dfa256 --len 4095 --rows 4096 --batch 524288 --sets 2 --steps 4
run "cfg3 regex23 1048576x2048     " --config regex23 --batch 1048576 --len 2047 --rows 2048 --steps 5
run "cfg4 headers3 32768x32768     " --config headers3 --batch 32768 --len 32767 --rows 32768 --steps 5
run "regex1 2097152x1024 sets 2    " --batch 2097152 --sets 2 --steps 4
