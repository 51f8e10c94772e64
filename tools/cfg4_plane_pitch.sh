#!/bin/bash
# cfg 4 (def-parallel kernel, D = 3): does the distance between the three record planes of a quad row (nb x 16 bytes: 512 KiB at 32768 strings) matter?
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; p=r.get("placement") or {}
print("ms/step %.4f frac %.3f | %s | best %s steps %s" % (d["ms_per_step"], r["frac"], r["kernel"][:50], p.get("best_gbs"), p.get("steps")))'
B="python3 bench.py --warmup 2 --no-cpu-baseline --no-pmc --no-verify --no-spread --config headers3 --len 32767 --rows 32768 --steps 5"
for b in 32768 30720 34816 36864 40960 49152 65536; do echo -n "headers3 batch $b x 32768: "; timeout 400 $B --batch $b 2>/dev/null | python3 -c "$P"; done
