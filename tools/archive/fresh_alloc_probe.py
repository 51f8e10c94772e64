#!/usr/bin/env python3
"""cfg 3 shape in a FRESH process: launch time by how records / masked rows are allocated (one variant per process).  (tools only)
usage: fresh_alloc_probe.py <rec> <msk>   each one of: torch | contig | contig+<MiB pad in front> | slab (both from one contiguous slab)"""
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
RB, MB = B * M * 2 * 4, B * M * 2
st = torch.empty(B, dtype=torch.int64, device=dev)
def make(kind, nbytes):
    if kind == "torch": return torch.empty(nbytes, dtype=torch.uint8, device=dev)
    pad = int(kind.split("+")[1]) << 20 if "+" in kind else 0
    return hra.DeviceBuffer(nbytes + pad, 0).tensor()[pad:]
rk, mk = sys.argv[1], sys.argv[2]
if rk.startswith("big"):      # records at the start of a contiguous allocation of <n> GiB, masked rows from torch
    keep = hra.DeviceBuffer(int(rk[3:]) << 30, 0).tensor(); rec, msk = keep[:RB], make(mk, MB)
elif rk == "slab":
    s = hra.DeviceBuffer(RB + MB, 0).tensor(); rec, msk = s[:RB], s[RB:]
else:
    rec, msk = make(rk, RB), make(mk, MB)
rec, msk = rec.view(torch.int32), msk.view(torch.int16)
def timeit(k=12):
    for _ in range(3): cfg.witness_batch_position_major(pm0, d_lens, out=(rec, msk, st), chars_pm_stride=2048)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): cfg.witness_batch_position_major(pm0, d_lens, out=(rec, msk, st), chars_pm_stride=2048)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
print("rec %-12s msk %-12s %7.1f %7.1f us   rec %#x msk %#x" % (rk, mk, timeit(), timeit(), rec.data_ptr(), msk.data_ptr()))
