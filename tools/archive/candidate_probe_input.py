#!/usr/bin/env python3
"""cfg 3 shape: records and masked rows from the library's pair search, then candidate INPUT buffers allocated one after the
other: does the placement of the read stream matter too?  (tools only)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import halo2_regex_amd as hra
from halo2_regex_amd import synth
B, N, M = 262144, 2047, 2048
D_DIR = os.path.join(ROOT, "tests", "golden", "dfa")
rd = lambda f: open(os.path.join(D_DIR, f), "rb").read()
defs = [hra.RegexDefs(hra.AllstrRegexDef(rd("regex%d_test_lookup.txt" % k)), [hra.SubstrRegexDef(rd("substr%d_test_lookup.txt" % k))]) for k in (2, 3)]
cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
dev = torch.device("cuda", 0)
chars, lens = synth.regex23_planted(B, N, seed=0, stride=2048)
d_lens = torch.from_numpy(lens.astype(np.int32)).to(dev)
pm0 = hra.chars_to_position_major(torch.from_numpy(chars).to(dev))
del chars
out = cfg.alloc_outputs_position_major(B, dev)
def timeit(pm, k=8):
    for _ in range(2): cfg.witness_batch_position_major(pm, d_lens, out=out, chars_pm_stride=2048)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): cfg.witness_batch_position_major(pm, d_lens, out=out, chars_pm_stride=2048)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
res = [timeit(pm0)]
keep = []
for c in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    pm = pm0.clone(); keep.append(pm)
    res.append(timeit(pm))
print("input candidates (0.5 GiB each, allocation order; the first is the original), us per launch: " + " ".join("%.0f" % x for x in res))
