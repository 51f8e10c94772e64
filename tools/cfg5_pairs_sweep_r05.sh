#!/bin/bash
# cfg 5 (dfa256, 131072 x 4096 B) against the number of tagged pairs, with one and with two substring definitions ("1-2 substr defs"):
# -> gpurun_out/r05_cfg5_pairs_sweep.txt  (pairs are PER definition)
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r05_cfg5_pairs; rm -rf $O; mkdir -p $O
for nd in 1 2; do for p in 2 20 100 200 2000 20000; do
  python3 bench.py --config dfa256 --batch 131072 --len 4095 --rows 4096 --steps 10 --warmup 3 --distinct 65536 --substr-pairs $p --substr-defs $nd --no-cpu-baseline --no-pmc > $O/d${nd}_p${p}.json 2>/dev/null
  python3 -c "
import json
d=json.loads(open('$O/d${nd}_p${p}.json').read().strip().splitlines()[-1]); r=d['roofline']; mc=r.get('mix_ceiling') or {}
print('substr defs %d  pairs each %-6d  %.4f ms  frac %.3f  kernel/pass %.2f  verified %s  %s' % ($nd, $p, r['avg_launch_ms'], r['frac'], mc.get('kernel_over_best_probe', 0), d['verified']['bit_exact'], r['kernel'][:64]))"
done; done | tee gpurun_out/r05_cfg5_pairs_sweep.txt
