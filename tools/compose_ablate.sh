#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
for dbg in 0 1 2 4 7; do for cfgargs in "--config headers3" "--config regex1"; do
rm -rf gpurun_out/ca_kt
HRX_SPEC_DBG=$dbg HRX_LIB_PATH=$PWD/halo2_regex_amd/csrc/libhrx_ablation.so rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ca_kt -o r1 -- python3 bench.py $cfgargs --batch 8192 --len 32767 --rows 32768 --sets 1 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-spread --no-verify --allow-debug-flags > /dev/null 2>&1
echo -n "dbg=$dbg $cfgargs: "; python3 - <<PY
import csv,glob
for f in glob.glob("gpurun_out/ca_kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "compose" in r["Name"] or "scout" in r["Name"]: print(r["Name"][5:22], "%.1f us" % (float(r["AverageNs"])/1e3), end="  ")
print()
PY
done; done; rm -rf gpurun_out/ca_kt
