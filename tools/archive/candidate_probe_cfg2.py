#!/usr/bin/env python3
"""The bench line's shape (regex1, 65536 x 1024 B: records 256 MiB, masked rows 128 MiB, Infinity-Cache regime when one buffer set is
re-written): launch time per masked-row candidate, the candidates being the first 128 MiB of consecutive 1-GiB allocations.  (tools only)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import halo2_regex_amd as hra
from halo2_regex_amd import synth
B, N, M = 65536, 1023, 1024
D_DIR = os.path.join(ROOT, "tests", "golden", "dfa")
rd = lambda f: open(os.path.join(D_DIR, f), "rb").read()
defs = [hra.RegexDefs(hra.AllstrRegexDef(rd("regex1_test_lookup.txt")), [hra.SubstrRegexDef(rd("substr1_test_lookup.txt"))])]
cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
dev = torch.device("cuda", 0)
chars, lens = synth.regex1_planted(B, N, seed=0, stride=1024)
d_lens = torch.from_numpy(lens.astype(np.int32)).to(dev)
pm0 = hra.chars_to_position_major(torch.from_numpy(chars).to(dev))
st = torch.empty(B, dtype=torch.int64, device=dev)
NS = 8
recs = [torch.empty(B * M, dtype=torch.int32, device=dev) for _ in range(NS)]
pms = [pm0.clone() for _ in range(NS)]
def timeit(rec, msk, k=100):
    for _ in range(5): cfg.witness_batch_position_major(pm0, d_lens, out=(rec, msk, st), chars_pm_stride=1024)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): cfg.witness_batch_position_major(pm0, d_lens, out=(rec, msk, st), chars_pm_stride=1024)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
def rotate(msks, k=160):
    for i in range(NS): cfg.witness_batch_position_major(pms[i], d_lens, out=(recs[i], msks[i], st), chars_pm_stride=1024)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for j in range(k):
        i = j % NS
        cfg.witness_batch_position_major(pms[i], d_lens, out=(recs[i], msks[i], st), chars_pm_stride=1024)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
bricks = [torch.empty(1 << 29, dtype=torch.int16, device=dev) for _ in range(40)]     # 1 GiB each
res = [timeit(recs[0], b[:B * M]) for b in bricks]
print("one buffer set re-written, masked rows = first 128 MiB of brick i: " + " ".join("%.1f" % x for x in res))
near = [torch.empty(B * M, dtype=torch.int16, device=dev) for _ in range(NS)]
print("8 sets in turn, masked rows allocated next to the records: %.1f us" % rotate(near))
for pick in (range(0, 8), range(8, 16), range(16, 24), range(24, 32), range(32, 40)):
    print("8 sets in turn, masked rows in bricks %d..%d: %.1f us" % (pick[0], pick[-1], rotate([bricks[i][:B * M] for i in pick])))
