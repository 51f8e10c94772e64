# cfg 2 string-major (regex1 65536 x 1024, hrx::witness_split_kernel<1, 32, false>): the launch without ... (ablation build; flags as in tools/sm_ablate_cfg5.sh)
one() { python bench.py --layout string-major --steps 100 --warmup 10 --no-other-configs --no-cpu-baseline --no-pmc --no-spread "$@" 2>&1 | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('%.4f ms  frac %.3f' % (l['ms_per_step'], l['roofline']['frac']))"; }
echo -n "release: "; one
echo -n "release, --dense: "; one --dense
export HRX_LIB_PATH=$PWD/halo2_regex_amd/csrc/libhrx_ablation.so
for f in 0 1 2 3 8 16 17 19 0x800000; do echo -n "ablation build, debug flags $f: "; HRX_DEBUG_FLAGS=$f one --allow-debug-flags --no-verify; done
