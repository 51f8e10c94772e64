#!/bin/bash
# String-major rows out of the def-parallel launch (hrx_kernel_pmd.hip SMO): what the launch costs without its record stores (1), without its masked-row stores (2),
# without the walkers' sub-tile writes (16), with every sub-tile stored onto the strings' first rows (0x1000000: the same instructions, no HBM behind them) — the ablation build, nothing verified.   tools/sm_ablate.sh [config] [rows]
cfg=${1:-headers5}; rows=${2:-1024}
one() { python bench.py --config $cfg --layout string-major --dense --rows $rows --len $((rows-1)) --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-pmc --no-spread "$@" 2>&1 | tail -1 |
  python -c "import sys,json; l=json.loads(sys.stdin.read()); print('%.4f ms  frac %.3f  %s' % (l['ms_per_step'], l['roofline']['frac'], l['config'].get('kernel','')))"; }
echo "release:"; one
export HRX_LIB_PATH=$PWD/halo2_regex_amd/csrc/libhrx_ablation.so
for f in 0 1 16 17 19 16777216; do echo "ablation build, debug flags $f:"; HRX_DEBUG_FLAGS=$f one --allow-debug-flags --no-verify; done
