#!/bin/bash
# the bench line with the driver's flags and with the defaults, and the arena walk's report: one lease per call
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; p=r.get("placement") or {}
print("ms/step %.4f value %.3e frac %.3f | steps %s chosen %s ref %s first %s best %s search_ms %s" % (d["ms_per_step"], d["value"], r["frac"], p.get("steps",[None])[0], p.get("chosen_step",[None])[0], p.get("ref_gbs",[None])[0], p.get("first_gbs",[None])[0], p.get("best_gbs",[None])[0], p.get("search_ms",[None])[0]))'
echo -n "K=20 W=5 : "; HRX_PLACE_TRACE=${TRACE:-0} timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-spread 2>gpurun_out/headline_trace.txt | python3 -c "$P"
echo -n "defaults : "; timeout 300 python3 bench.py --no-cpu-baseline --no-pmc --no-spread 2>/dev/null | python3 -c "$P"
grep "hrx placement: step" gpurun_out/headline_trace.txt | awk '{print $8}' | tr '\n' ' '; echo
