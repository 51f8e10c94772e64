#!/usr/bin/env python3
"""What hrx_alloc_outputs_position_major measures (tools only; HRX_LIB_PATH=.../libhrx_ablation.so HRX_PLACE_TRACE=1): the probe's
microseconds per masked-row candidate, and the launch time with the pair it kept.  usage: place_trace.py B M config"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import halo2_regex_amd as hra
from halo2_regex_amd import synth
B, M, which = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
D_DIR = os.path.join(ROOT, "tests", "golden", "dfa")
rd = lambda f: open(os.path.join(D_DIR, f), "rb").read()
mk = lambda k: hra.RegexDefs(hra.AllstrRegexDef(rd("regex%d_test_lookup.txt" % k)), [hra.SubstrRegexDef(rd("substr%d_test_lookup.txt" % k))])
hdr = lambda n, ns: hra.RegexDefs(hra.AllstrRegexDef(rd(n + "_lookup.txt")), [hra.SubstrRegexDef(rd("%s_substr%d.txt" % (n, k))) for k in range(ns)])
if which == "regex23": defs, gen = [mk(2), mk(3)], synth.regex23_planted
elif which == "headers3": defs, gen = [hdr("header_from", 1), hdr("header_to", 1), hdr("header_subject", 3)], synth.headers_planted
else: defs, gen = [mk(1)], synth.regex1_planted
cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
dev = torch.device("cuda", 0)
chars, lens = gen(B, M - 1, seed=0, stride=M)
d_lens = torch.from_numpy(lens.astype(np.int32)).to(dev)
pm0 = hra.chars_to_position_major(torch.from_numpy(chars).to(dev))
del chars
out = cfg.alloc_outputs_position_major(B, dev)
def timeit(k=8):
    for _ in range(2): cfg.witness_batch_position_major(pm0, d_lens, out=out, chars_pm_stride=M)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): cfg.witness_batch_position_major(pm0, d_lens, out=out, chars_pm_stride=M)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
sys.stderr.flush()
print("%s %d x %d: %.1f us per launch with the pair kept (records %#x, masked %#x)" % (which, B, M, timeit(), out[0].data_ptr(), out[1].data_ptr()))
