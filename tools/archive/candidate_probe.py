#!/usr/bin/env python3
"""cfg 3 shape: ONE records buffer, candidate masked-row buffers allocated further and further away (spacer allocations held in
between): launch time per candidate.  How far does the allocator have to go for a masked-row buffer that does not collide with
the records?  (tools only)   usage: candidate_probe.py [spacer GiB = 8] [candidates = 14] [batch = 262144]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import halo2_regex_amd as hra
from halo2_regex_amd import synth
SP = float(sys.argv[1]) if len(sys.argv) > 1 else 8
NC = int(sys.argv[2]) if len(sys.argv) > 2 else 14
B = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
N, M = 2047, 2048
D_DIR = os.path.join(ROOT, "tests", "golden", "dfa")
rd = lambda f: open(os.path.join(D_DIR, f), "rb").read()
defs = [hra.RegexDefs(hra.AllstrRegexDef(rd("regex%d_test_lookup.txt" % k)), [hra.SubstrRegexDef(rd("substr%d_test_lookup.txt" % k))]) for k in (2, 3)]
cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
dev = torch.device("cuda", 0)
chars, lens = synth.regex23_planted(B, N, seed=0, stride=2048)
d_lens = torch.from_numpy(lens.astype(np.int32)).to(dev)
pm0 = hra.chars_to_position_major(torch.from_numpy(chars).to(dev))
del chars
st = torch.empty(B, dtype=torch.int64, device=dev)
rec = torch.empty(B * M * 2, dtype=torch.int32, device=dev)
def timeit(msk, k=6):
    for _ in range(2): cfg.witness_batch_position_major(pm0, d_lens, out=(rec, msk, st), chars_pm_stride=2048)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): cfg.witness_batch_position_major(pm0, d_lens, out=(rec, msk, st), chars_pm_stride=2048)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
keep, res = [], []
t0 = time.time()
for c in range(NC):
    msk = torch.empty(B * M, dtype=torch.int16, device=dev)
    res.append(timeit(msk))
    keep.append(msk)
    if SP > 0:
        try: keep.append(torch.empty(int(SP * (1 << 30)), dtype=torch.uint8, device=dev))
        except RuntimeError: break
torch.cuda.synchronize()
print("spacer %.0f GiB, masked-row candidates in allocation order, us per launch (%.1f s in all): " % (SP, time.time() - t0) + " ".join("%.0f" % x for x in res))
