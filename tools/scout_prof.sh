cd /tmp; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for nc in 1 0; do for cfg in "regex1 8192" "headers3 8192"; do set -- $cfg
  P=$R/gpurun_out/prof_scout_${1}_nc$nc; rm -rf $P; mkdir -p $P
  if [ $nc = 1 ]; then export HRX_SPEC_NO_COMPACT=1; else unset HRX_SPEC_NO_COMPACT; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o r1 -- python3 bench.py --config $1 --batch $2 --len 32767 --rows 32768 --steps 10 --warmup 2 --distinct 2048 --no-cpu-baseline --no-pmc --no-spread --no-verify > $P/log 2>&1
  echo "== no_compact=$nc $1 $2"; python3 - <<PY
import csv,glob
f=glob.glob("$P/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("spec_","witness_")): print("   %-60s calls %4s avg %8.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
done; done
