# hrx_witness_batch_host inside the bench process over a few processes (HRX_HOST_TRACE=1: one line per call in gpurun_out/e2e_*.err): the shipped choice (the context measures both ways)
# against always pipelined (HRX_HOST_PIPELINE=1: 8.0 ms on some boxes of the pool, 16.5 on others) and always one stream (=0)
run() { HRX_HOST_TRACE=1 python bench.py --no-other-configs --no-pmc --no-cpu-baseline --steps 20 --warmup 5 --no-verify 2> gpurun_out/e2e_$1.err | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); e=l['end_to_end_host']; print('$1: ms_per_call %.2f min %.2f copy alone %.2f status ok %s  comparison %s' % (e['ms_per_call'], e['ms_min'], e['copy_out_alone_ms'], e['status_ok'], {k: ['%.1f' % x for x in v] for k, v in e['comparison_calls_ms'].items()}))"; }
for i in 1 2 3; do
  run measured_$i
  HRX_HOST_PIPELINE=1 run pipelined_$i
  HRX_HOST_PIPELINE=0 run one_stream_$i
done
grep -h "hrx host" gpurun_out/e2e_measured_1.err | cut -c1-160 | grep -v "chunks of"
