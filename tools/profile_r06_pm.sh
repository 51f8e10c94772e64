#!/bin/bash
# The rocprofv3 passes behind profiles/r06_pm_*: run on the GPU box from the repo root (gpurun -- 'bash tools/profile_r06_pm.sh').
# The bench line in its graded regime (the timed steps rotate over 8 buffer sets): kernel trace + stats in one pass (only the rotating
# launches: --no-spread keeps the one-buffer-set and no-compute probes out of the trace), PMC counters in their own passes, csv output;
# tools/summarize_prof.py condenses gpurun_out/prof into gpurun_out/r06_pm_{kernel_stats.csv,pmc.json}; then the untraced default line.
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; P=$R/gpurun_out/prof; rm -rf $P; mkdir -p $P; cd $R
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt -o r1 -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pmc --no-verify --no-spread > $P/kt.log 2>&1; echo kt rc=$?
python3 tools/trace_replays.py $P/kt --kernel witness_pm > $R/gpurun_out/r06_pm_trace_replays.txt; cat $R/gpurun_out/r06_pm_trace_replays.txt
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch -o r1 -- python3 bench.py --eager --steps 5 --warmup 1 --sets 1 --no-cpu-baseline --no-verify --no-spread --no-pmc > $P/fetch.log 2>&1; echo fetch rc=$?
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write -o r1 -- python3 bench.py --eager --steps 5 --warmup 1 --sets 1 --no-cpu-baseline --no-verify --no-spread --no-pmc > $P/write.log 2>&1; echo write rc=$?
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $P/sq -o r1 -- python3 bench.py --eager --steps 5 --warmup 1 --sets 1 --no-cpu-baseline --no-verify --no-spread --no-pmc > $P/sq.log 2>&1; echo sq rc=$?
python3 tools/summarize_prof.py $P $R/gpurun_out/r06_pm > /dev/null && echo summarized
grep '^{' $P/kt.log > $R/gpurun_out/r06_bench_line_traced.json   # the traced process's own bench line: its HIP-event figure belongs to the same process as the kernel stats
timeout 400 python3 bench.py > $R/gpurun_out/r06_bench_line.json 2> $R/gpurun_out/r06_bench_line.err; tail -c 900 $R/gpurun_out/r06_bench_line.json
timeout 400 python3 bench.py --steps 20 --warmup 5 > $R/gpurun_out/r06_bench_line_driver_flags.json 2>> $R/gpurun_out/r06_bench_line.err
python3 - <<PY
import json
for f in ("r06_bench_line", "r06_bench_line_driver_flags", "r06_bench_line_traced"):
    d = json.loads(open("$R/gpurun_out/%s.json" % f).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f, "ms/step %.4f frac %.3f avg_launch %.4f" % (d["ms_per_step"], r["frac"], r["avg_launch_ms"]), "one_set", (r.get("one_buffer_set") or {}).get("frac"), "probe", (r.get("mix_ceiling") or {}).get("traffic_pass_us"),
          "k/probe", (r.get("mix_ceiling") or {}).get("kernel_over_best_probe"), "traffic", r.get("traffic"), "verified", (d.get("verified") or {}).get("strings"))
p = json.load(open("$R/gpurun_out/r06_pm_pmc.json"))
k = [x for x in p["kernel_stats"] if "witness" in x["name"]][0]
print("kernel stats: %s avg %.2f us over %d calls" % (k["name"][:60], k["avg_ns"] / 1e3, k["calls"]), "hbm", p.get("hbm_bytes_per_launch"))
PY
