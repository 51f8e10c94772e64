#!/bin/bash
# One lease of the pool = one line: the default bench run with the driver's flags -> gpurun_out/r06_leases/<tag>.json + one summary line (headline, the three legs, the host routes).
# Called once per gpurun call (every call gets a fresh box): tools/lease_line.sh a, ... b, ...; profiles/r06_leases/summary.txt collects the summary lines.
cd ${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-a}; mkdir -p gpurun_out/r06_leases
timeout 900 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r06_leases/$tag.json 2> gpurun_out/r06_leases/$tag.err
python3 - <<PY
import json
l = json.loads(open("gpurun_out/r06_leases/$tag.json").read().strip().splitlines()[-1]); r = l["roofline"]
legs = " ".join("%s %.3f (interleaved %s)" % (o["baseline_config"].split(":")[0], o["frac"], ("%.3f" % o["interleaved_layout"]["frac"]) if o.get("interleaved_layout") else "-") for o in l.get("other_configs", []) if "frac" in o)
e = (l.get("end_to_end_host") or {}).get("routes") or {}
print("lease $tag: headline %.3f (pass %.1f us, plain allocations %s) | %s | host call auto %.2f ms (%s) device %.2f host %.2f | verified %s" % (
    r["frac"], (r.get("mix_ceiling") or {}).get("traffic_pass_us", 0), ("%.3f" % r["plain_allocations"]["frac"]) if r.get("plain_allocations") else "-", legs,
    e.get("auto_ms", 0), e.get("auto_chose"), e.get("device_ms", 0), e.get("host_walk_all_cores_ms", 0), (l.get("verified") or {}).get("bit_exact")))
PY
