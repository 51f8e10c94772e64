#!/usr/bin/env python3
"""Condense a rocprofv3 run directory (csv output) into the small files kept under profiles/.

  tools/summarize_prof.py gpurun_out/prof profiles/r01   ->  profiles/r01_kernel_stats.csv, profiles/r01_pmc.json

PMC units follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE counts 128-byte read requests at 64 B, so it is doubled before it is compared with a byte count.
"""
import collections
import csv
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
shutil.copyfile(os.path.join(src, "kt", "r1_kernel_stats.csv"), dst + "_kernel_stats.csv")
out = {"kernel_stats": [], "pmc_per_launch": {}, "notes": "separate rocprofv3 --pmc passes, one counter group each; means over the launches of the pass"}
for r in csv.DictReader(open(os.path.join(src, "kt", "r1_kernel_stats.csv"))):
    out["kernel_stats"].append({"name": r["Name"], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])})
for name in ("fetch", "write", "sq"):
    f = os.path.join(src, name, "r1_counter_collection.csv")
    if not os.path.exists(f):
        continue
    agg = collections.defaultdict(list)
    kern = None
    for r in csv.DictReader(open(f)):
        if "witness" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            kern = r["Kernel_Name"]
            out["launch"] = {"grid": r["Grid_Size"], "workgroup": r["Workgroup_Size"], "lds": r["LDS_Block_Size"],
                             "vgpr": r["VGPR_Count"], "sgpr": r["SGPR_Count"]}
    for k, v in agg.items():
        out["pmc_per_launch"][k] = sum(v) / len(v)
    out["kernel"] = kern
    # every kernel of the library (a chunked launch is scout + compose + walk + stitch + repair; string-major via the transpose kernel)
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "hrx::" in r["Kernel_Name"] and "placement_probe" not in r["Kernel_Name"] and "traffic_pass" not in r["Kernel_Name"]:
            by[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn, cs in by.items():
        for c, v in cs.items():
            out.setdefault("pmc_by_kernel", {}).setdefault(kn, {})[c] = sum(v) / len(v)
p = out["pmc_per_launch"]
if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
    out["hbm_bytes_per_launch"] = {
        "read": p["FETCH_SIZE"] * 1024 * 2,       # gfx950 correction: x2
        "written": p["WRITE_SIZE"] * 1024,
    }
    out["hbm_bytes_per_launch"]["total"] = out["hbm_bytes_per_launch"]["read"] + out["hbm_bytes_per_launch"]["written"]
json.dump(out, open(dst + "_pmc.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
