#!/usr/bin/env python3
"""cfg 3 shape: which OUTPUT buffer carries the fast / slow state, records or masked rows?  (tools only)"""
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
r0, m0, st = cfg.alloc_outputs_position_major(B, dev)
def timeit(pm, out, k=12):
    for _ in range(2): cfg.witness_batch_position_major(pm, d_lens, out=out, chars_pm_stride=2048)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): cfg.witness_batch_position_major(pm, d_lens, out=out, chars_pm_stride=2048)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
NS = 8
recs = [torch.empty_like(r0) for _ in range(NS)]
msks = [torch.empty_like(m0) for _ in range(NS)]
print("free/total GiB: %.1f / %.1f" % tuple(x / 2**30 for x in torch.cuda.mem_get_info()))
print("        " + " ".join("msk%d   " % j for j in range(NS)))
for i in range(NS):
    print("rec%d  " % i + " ".join("%7.1f" % timeit(pm0, (recs[i], msks[j], st)) for j in range(NS)), flush=True)
print("rec ptrs " + " ".join("%#x" % r.data_ptr() for r in recs))
print("msk ptrs " + " ".join("%#x" % m.data_ptr() for m in msks))
# halves: does the state belong to a part of the records buffer?  (131072 strings = 2 blocks = half of every buffer)
