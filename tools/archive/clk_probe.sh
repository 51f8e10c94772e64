#!/bin/bash
# Per-process launch time of the cfg 3 shape next to the DPM clocks / power sampled from sysfs while it runs.
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r02/clk_probe.txt; mkdir -p gpurun_out/r02; : > $O
D=$(ls -d /sys/class/drm/card*/device | head -1); ls $D | tr '\n' ' ' >> $O; echo >> $O
H=$(ls -d $D/hwmon/hwmon* | head -1); ls $H | tr '\n' ' ' >> $O; echo >> $O
B="python bench.py --no-cpu-baseline --no-pmc --no-verify --no-spread --config regex23 --batch 262144 --len 2047 --rows 2048 --steps 400 --warmup 3"
for i in 1 2 3 4 5 6; do
  ( while true; do
      s=""; for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk pp_dpm_socclk; do s="$s $f=$(grep '\*' $D/$f 2>/dev/null | tr -d '\n')"; done
      echo "$s pw=$(cat $H/power1_average 2>/dev/null || cat $H/power1_input 2>/dev/null) t=$(cat $H/temp1_input 2>/dev/null) busy=$(cat $D/gpu_busy_percent 2>/dev/null) mem=$(cat $D/mem_busy_percent 2>/dev/null)"; sleep 0.1; done ) > /tmp/clk_$i.txt 2>&1 &
  S=$!
  $B | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run $i %8.3f ms frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))" >> $O
  kill $S; wait $S 2>/dev/null
  sort /tmp/clk_$i.txt | uniq -c | sort -rn | head -8 >> $O
done
cat $O
