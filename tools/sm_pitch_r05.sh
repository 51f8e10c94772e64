one() { python bench.py "$@" --layout string-major --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-pmc --no-spread 2>&1 | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('%.4f ms  frac %.3f  %s' % (l['ms_per_step'], l['roofline']['frac'], {k:v for k,v in l['config'].items() if 'pitch' in k or 'kernel' in k}))"; }
for c in headers4 headers5; do for r in 1024 2048; do
echo -n "$c x $r dense: "; one --config $c --rows $r --len $((r-1)) --dense
echo -n "$c x $r recommended pitches: "; one --config $c --rows $r --len $((r-1))
done; done
