#!/bin/bash
# how many untimed replays of the K-step graph the timed replay needs behind it (profiles/r03_probes/warmup_ramp.txt)
cd "$(dirname "$0")/.." || exit 1
for i in 1 2; do for k in 20 40 100 200; do for u in 5 50; do
echo -n "K=$k untimed replays=$u: "; python3 bench.py --steps $k --warmup 5 --untimed-replays $u --no-cpu-baseline --no-pmc --no-spread --no-verify 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step %.4f (events %.4f) frac %.3f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
done; done; done
