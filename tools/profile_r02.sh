#!/bin/bash
# The rocprofv3 passes behind profiles/r02_pm_*: run on the GPU box from the repo root (gpurun -- 'bash tools/profile_r02.sh').
# Kernel trace + stats in one pass; PMC counters in their own passes (one group each), csv output.
# tools/summarize_prof.py condenses gpurun_out/prof into gpurun_out/r02_pm_{kernel_stats.csv,pmc.json}.
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; P=$R/gpurun_out/prof; rm -rf $P; mkdir -p $P; cd $R
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt -o r1 -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pmc --no-verify --no-spread > $P/kt.log 2>&1; echo kt rc=$?
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch -o r1 -- python3 bench.py --eager --steps 5 --warmup 1 --no-cpu-baseline --no-verify --no-spread --no-pmc --no-verify --no-spread > $P/fetch.log 2>&1; echo fetch rc=$?
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write -o r1 -- python3 bench.py --eager --steps 5 --warmup 1 --no-cpu-baseline --no-verify --no-spread --no-pmc --no-verify --no-spread > $P/write.log 2>&1; echo write rc=$?
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $P/sq -o r1 -- python3 bench.py --eager --steps 5 --warmup 1 --no-cpu-baseline --no-verify --no-spread --no-pmc --no-verify --no-spread > $P/sq.log 2>&1; echo sq rc=$?
python3 tools/summarize_prof.py $P $R/gpurun_out/r02_pm > /dev/null && echo summarized
grep '^{' $P/kt.log > $R/gpurun_out/r02_bench_line_traced.json   # the traced process's own bench line: its HIP-event figure belongs to the same process as the kernel stats
timeout 400 python3 bench.py > $R/gpurun_out/r02_bench_line.json 2> $R/gpurun_out/r02_bench_line.err; tail -c 600 $R/gpurun_out/r02_bench_line.json
