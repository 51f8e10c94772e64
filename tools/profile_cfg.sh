#!/bin/bash
# rocprofv3 passes for any bench configuration: tools/profile_cfg.sh <tag> <bench args...> -> gpurun_out/<tag>_{kernel_stats.csv,pmc.json}
# (kernel trace + stats in one pass; FETCH_SIZE / WRITE_SIZE in their own passes; copy what is cited into profiles/)
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; shift; P=$R/gpurun_out/prof_$tag; rm -rf $P; mkdir -p $P; cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt -o r1 -- python3 bench.py --no-cpu-baseline --no-pmc --no-spread "$@" > $P/kt.log 2>&1; echo kt rc=$?
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch -o r1 -- python3 bench.py --eager --steps 3 --warmup 1 --sets 1 --no-cpu-baseline --no-verify --no-spread --no-pmc "$@" > $P/fetch.log 2>&1; echo fetch rc=$?
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write -o r1 -- python3 bench.py --eager --steps 3 --warmup 1 --sets 1 --no-cpu-baseline --no-verify --no-spread --no-pmc "$@" > $P/write.log 2>&1; echo write rc=$?
python3 tools/summarize_prof.py $P $R/gpurun_out/$tag > /dev/null && echo summarized
python3 - <<PY
import json
d = json.load(open("$R/gpurun_out/${tag}_pmc.json"))
line = json.loads([l for l in open("$P/kt.log").read().split("\n") if l.startswith("{")][-1])
k = [x for x in d["kernel_stats"] if "witness" in x["name"]][0]
h = d.get("hbm_bytes_per_launch", {})
print("%s: kernel avg %.1f us over %d calls (bench: %.1f us); algorithmic %.1f MB, HBM read %.1f MB + written %.1f MB = %.1f MB"
      % ("$tag", k["avg_ns"] / 1e3, k["calls"], line["roofline"]["avg_launch_ms"] * 1e3, line["roofline"]["algorithmic_bytes_per_launch"] / 1e6,
         h.get("read", 0) / 1e6, h.get("written", 0) / 1e6, h.get("total", 0) / 1e6))
PY
