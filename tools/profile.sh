#!/bin/bash
# The rocprofv3 passes behind profiles/r01_pm_*: run on the GPU box from the repo root (gpurun -- 'bash tools/profile.sh').
# Kernel trace + stats in one pass; PMC counters in their own passes (one group each), csv output (the default rocpd
# output + --stats hung on this pool).  tools/summarize_prof.py condenses gpurun_out/prof into profiles/.
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; P=$R/gpurun_out/prof; rm -rf $P; mkdir -p $P; cd $R
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt -o r1 -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pmc > $P/kt.log 2>&1; echo kt rc=$?
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch -o r1 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-verify --no-spread --no-pmc > $P/fetch.log 2>&1; echo fetch rc=$?
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write -o r1 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-verify --no-spread --no-pmc > $P/write.log 2>&1; echo write rc=$?
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $P/sq -o r1 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-verify --no-spread --no-pmc > $P/sq.log 2>&1; echo sq rc=$?
python3 tools/summarize_prof.py $P $R/gpurun_out/r01_pm > /dev/null && echo summarized
timeout 300 python bench.py > $R/gpurun_out/bench_final.json 2> $R/gpurun_out/bench_final.err; tail -c 400 $R/gpurun_out/bench_final.json
timeout 200 python3 tools/host_path_rate.py > $R/gpurun_out/host_path.txt 2>&1; cat $R/gpurun_out/host_path.txt
