#!/bin/bash
# power / clock readings (rocm-smi, 4 samples) while a sustained load runs: tools/smi_during.sh <load 0|1> [gap_us]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
tools/sustain_clk 1500 40 ${2:-0} $1 > /tmp/sc_$1.txt 2>&1 &
pid=$!
sleep 1.5
for i in 1 2 3 4; do
  rocm-smi --showpower --showclocks --showuse --showmemuse 2>&1 | grep -E "fclk|mclk|sclk|socclk|Power \(W\)|busy|GPU use" | tr '\n' ';'; echo
  sleep 0.4
done
wait $pid
sed -n '1p;200p;600p;1000p;1400p' /tmp/sc_$1.txt
