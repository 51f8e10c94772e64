#!/bin/bash
# SQ counter passes for one bench configuration (profiling only): tools/pmc_cfg.sh <tag> <bench args...>
cd /tmp; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
tag=$1; shift; P=gpurun_out/prof_$tag; rm -rf $P; mkdir -p $P
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $P/sq -o r1 -- python3 bench.py --eager --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-spread --no-pmc "$@" > $P/sq.log 2>&1; echo sq rc=$?
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $P/sq2 -o r1 -- python3 bench.py --eager --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-spread --no-pmc "$@" > $P/sq2.log 2>&1; echo sq2 rc=$?
python3 - <<PY
import csv,collections
for d in ("sq","sq2"):
    agg=collections.defaultdict(list)
    try:
        for r in csv.DictReader(open("$P/%s/r1_counter_collection.csv"%d)):
            if "witness" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    except Exception as e: print(d, e)
    for k,v in agg.items(): print("%-24s %.4g" % (k, sum(v)/len(v)))
PY
tail -3 $P/sq2.log | cut -c1-300
