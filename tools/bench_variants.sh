#!/bin/bash
# quick on-box A/B: the bench line under the profiling/ablation flags HRX_DEBUG_FLAGS (not a product path)
for f in ${FLAGS:-0 1 2 3}; do
  HRX_DEBUG_FLAGS=$f timeout 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline ${BENCH_EXTRA} 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('flags=$f', 'ms/step=%.4f'%j['ms_per_step'], 'rows/s=%.3e'%j['value'], 'frac=%.3f'%j['roofline']['frac'])"
done
