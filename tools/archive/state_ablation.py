#!/usr/bin/env python3
"""cfg 3 shape: a FAST and a SLOW records buffer of one process under the ablation bits (libhrx_ablation.so re-reads
HRX_DEBUG_FLAGS at every launch): which part of the launch's traffic does the slow placement hurt?  (tools only)
run with HRX_LIB_PATH=halo2_regex_amd/csrc/libhrx_ablation.so"""
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
m0 = torch.empty(B * M, dtype=torch.int16, device=dev); st = torch.empty(B, dtype=torch.int64, device=dev)
def timeit(rec, k=10):
    for _ in range(2): cfg.witness_batch_position_major(pm0, d_lens, out=(rec, m0, st), chars_pm_stride=2048)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): cfg.witness_batch_position_major(pm0, d_lens, out=(rec, m0, st), chars_pm_stride=2048)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
os.environ["HRX_DEBUG_FLAGS"] = "0"
recs = [torch.empty(B * M * 2, dtype=torch.int32, device=dev) for _ in range(10)]
t = [timeit(r) for r in recs]
print("as shipped: " + " ".join("%.0f" % x for x in t))
fast, slow = recs[int(np.argmin(t))], recs[int(np.argmax(t))]
for name, flags in (("as shipped", 0), ("no masked-row stores", 2), ("input from L2 (no HBM reads)", 4), ("no masked, no reads", 6), ("no record stores", 1),
                    ("write-back record stores", 32), ("write-back masked stores", 64), ("all write-back", 96), ("static groups", 1 << 11), ("dynamic groups", 1 << 12),
                    ("narrow table", 0x80000), ("def-parallel kernel", 0x4000000)):
    os.environ["HRX_DEBUG_FLAGS"] = hex(flags)
    print("%-30s fast buffer %7.1f   slow buffer %7.1f us" % (name, timeit(fast), timeit(slow)), flush=True)
