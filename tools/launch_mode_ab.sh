#!/bin/bash
# graph replay vs eager launches of the K timed steps, at the driver's K = 20 and at the default K = 200
cd "$(dirname "$0")/.." || exit 1
for i in 1 2 3; do for mode in "" "--eager"; do for k in 20 200; do
echo -n "steps=$k ${mode:-graph}: "; python3 bench.py --steps $k --warmup 5 $mode --no-cpu-baseline --no-pmc --no-spread 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step %.4f (events %.4f) frac %.3f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
done; done; done
