#!/bin/bash
# per-launch footprint sweep: does a launch over > 8 GB of buffers run slower per row?  (cfg 5 and the narrow-table regex1 kernel, with the no-compute pass beside each)
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; mc=d.get("memory_ceiling") or {}
print("ms/step %.4f frac %.3f probe_ms %s  %s %s" % (d["ms_per_step"], r["frac"], mc.get("traffic_pass_us") and "%.4f" % (mc["traffic_pass_us"]/1e3), r["kernel"][:60], r["launch"]))'
B="python3 bench.py --warmup 2 --no-cpu-baseline --no-pmc --no-verify"
for bs in "262144 2 6" "393216 2 4" "524288 2 4"; do set -- $bs
echo -n "dfa256 batch $1 x 4096: "; timeout 300 $B --config dfa256 --len 4095 --rows 4096 --batch $1 --sets $2 --steps $3 2>/dev/null | python3 -c "$P"; done
for bs in "1048576 2 6" "2097152 2 4"; do set -- $bs
echo -n "regex1 batch $1 x 1024: "; timeout 300 $B --batch $1 --sets $2 --steps $3 2>/dev/null | python3 -c "$P"; done
for bs in "262144 2 6" "524288 2 4"; do set -- $bs
echo -n "regex1 batch $1 x 4096: "; timeout 300 $B --len 4095 --rows 4096 --batch $1 --sets $2 --steps $3 2>/dev/null | python3 -c "$P"; done
