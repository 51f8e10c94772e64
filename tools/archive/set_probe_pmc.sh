#!/bin/bash
# set_probe.py under rocprofv3 --pmc: UTCL1 (per-CU TLB) requests / misses per launch, per buffer set.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02/set_probe_pmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PROBE_K=2 PROBE_ROUNDS=1
for pass in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum GRBM_UTCL2_BUSY" "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS TCP_UTCL1_LFIFO_FULL"; do
  d=$O/$(echo $pass | tr ' ' '+'); 
  rocprofv3 --pmc $pass -d $d -o p --output-format csv -- python3 $R/tools/set_probe.py 6 > $d.txt 2>&1
  python3 - "$d" <<'PY'
import sys, csv, glob, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
per = collections.OrderedDict()
for r in rows:
    if "witness" not in r["Kernel_Name"]: continue
    per.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
disp = list(per.values())
print(d.split("/")[-1], "dispatches", len(disp))
for s in range(0, len(disp), 5):
    grp = disp[s:s + 5]
    print("  set %d: " % (s // 5) + "  ".join("%s=%.4g" % (k, sum(g[k] for g in grp) / len(grp)) for k in grp[0]))
PY
  grep round $d.txt
done
