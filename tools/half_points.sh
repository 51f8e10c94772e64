cd ${GRAFT_REPO_ROOT:-/root/repo}
for spec in "1 400" "1 2000" "2 200" "2 2000" "1 200"; do set -- $spec
  python3 bench.py --config dfa256 --batch 131072 --len 4095 --rows 4096 --steps 10 --warmup 3 --distinct 65536 --substr-pairs $2 --substr-defs $1 --no-cpu-baseline --no-pmc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; mc=r.get('mix_ceiling') or {}
print('substr defs $1  pairs each %-6s  %.4f ms  frac %.3f  kernel/pass %.2f  verified %s  %s' % ('$2', r['avg_launch_ms'], r['frac'], mc.get('kernel_over_best_probe', 0), d['verified']['bit_exact'], r['kernel'][:64]))"
done
