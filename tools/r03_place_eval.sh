#!/bin/bash
# round 3: the bench line in the rotating-buffer regime, with and without the placement search, fresh processes alternating
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_place; mkdir -p $O
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc"
for i in 1 2 3; do
  HRX_PLACE=0 timeout 300 $B > $O/plain_$i.json 2> $O/plain_$i.err
  HRX_PLACE_TRACE=1 timeout 300 $B > $O/placed_$i.json 2> $O/placed_$i.err
done


python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03_place/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "FAILED", e); continue
    r = d["roofline"]
    print(f.split("/")[-1], "ms/step %.4f frac %.3f" % (d["ms_per_step"], r["frac"]), "one_set", r.get("one_buffer_set", {}).get("ms_per_step_median"),
          "probe", (r.get("mix_ceiling") or {}).get("traffic_pass_us"), "k/probe", (r.get("mix_ceiling") or {}).get("kernel_over_best_probe"),
          "verified", d.get("verified", {}).get("bit_exact"), d.get("verified", {}).get("strings"), "placement", {k: v for k, v in (r.get("placement") or {}).items() if k != "what"})
PY
