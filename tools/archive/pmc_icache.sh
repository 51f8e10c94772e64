#!/bin/bash
# instruction-cache counters of the bench-line kernel, release and ablation builds (profiling only)
cd /tmp; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for lib in libhrx.so libhrx_ablation.so; do
  P=gpurun_out/prof_ic_$lib; rm -rf $P; mkdir -p $P
  HRX_LIB_PATH=$R/halo2_regex_amd/csrc/$lib timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $P -o r1 -- python3 bench.py --eager --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-verify --no-spread --allow-debug-flags > $P/log.txt 2>&1; echo "$lib rc=$?"
  python3 - <<PY
import csv,collections,glob
agg=collections.defaultdict(list)
for f in glob.glob("$P/**/r1_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "witness" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print("  %-22s %.4g" % (k, sum(v)/len(v)))
PY
  tail -2 $P/log.txt | cut -c1-200
done
