one() { python bench.py --config dfa256 --batch 65536 --rows 4096 --len 4095 --layout string-major --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-pmc --no-spread "$@" 2>&1 | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('%.4f ms  frac %.3f' % (l['ms_per_step'], l['roofline']['frac']))"; }
echo -n "release: "; one
export HRX_LIB_PATH=$PWD/halo2_regex_amd/csrc/libhrx_ablation.so
for f in 0 1 2 3 8 16 19 0x800000 0x800003; do echo -n "ablation build, debug flags $f: "; HRX_DEBUG_FLAGS=$f one --allow-debug-flags --no-verify; done
