#!/bin/bash
# cfg 5 against the number of tagged (state, next) pairs of its substring definition: how much of the launch is the reveal-mask fix-ups (repairs of masked rows already stored)?
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]
print("ms/step %.4f frac %.3f verified %s" % (d["ms_per_step"], r["frac"], (d.get("verified") or {}).get("bit_exact")))'
B="python3 bench.py --config dfa256 --len 4095 --rows 4096 --warmup 3 --no-cpu-baseline --no-pmc --no-spread --batch 131072 --steps 10"
for p in 0 2 20 200 2000 20000; do echo -n "substr pairs $p: "; timeout 300 $B --substr-pairs $p 2>/dev/null | python3 -c "$P"; done
echo -n "substr pairs 200, fix-ups skipped (wrong output): "; HRX_DEBUG_FLAGS=$(python3 -c "
import re
s=open('halo2_regex_amd/csrc/hrx_kernel.hpp').read(); m=re.search(r'kDbgSkipFixups\s*=\s*([0-9a-fx<u ]+)',s); print(m.group(1).strip())" | python3 -c "import sys; e=sys.stdin.read().strip().replace('u',''); print(hex(eval(e)))") timeout 300 $B --allow-debug-flags --no-verify 2>/dev/null | python3 -c "$P"
