#!/usr/bin/env python3
"""cfg 3 shape (4 position-major blocks, 4-GiB records): launch time with the records at consecutive offsets of ONE physically
contiguous slab (hrx_device_alloc), masked rows and input fixed.  (tools only)"""
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
GIB = 1 << 30
RB = 4 * GIB
m0 = torch.empty(B * M, dtype=torch.int16, device=dev); st = torch.empty(B, dtype=torch.int64, device=dev)
slab = hra.DeviceBuffer(int(os.environ.get("SLAB_GIB", "40")) * GIB, 0).tensor()
print("slab %#x" % slab.data_ptr())
def timeit(rec, k=8):
    for _ in range(2): cfg.witness_batch_position_major(pm0, d_lens, out=(rec, m0, st), chars_pm_stride=2048)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): cfg.witness_batch_position_major(pm0, d_lens, out=(rec, m0, st), chars_pm_stride=2048)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
def timeit2(rec, msk, k=8):
    for _ in range(1): cfg.witness_batch_position_major(pm0, d_lens, out=(rec, msk, st), chars_pm_stride=2048)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): cfg.witness_batch_position_major(pm0, d_lens, out=(rec, msk, st), chars_pm_stride=2048)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
if os.environ.get("MAP2D"):
    # 2-D map: records at r GiB, masked rows at m GiB of the slab
    S = slab.numel() // GIB
    step = int(os.environ.get("MAP2D"))
    print("columns: masked rows at GiB " + " ".join("%4d" % m for m in range(0, S, step)))
    for r in range(0, S - 3, int(os.environ.get("RSTEP", "16"))):
        rec = slab[r * GIB: r * GIB + RB].view(torch.int32)
        row = []
        for m in range(0, S, step):
            row.append("   -" if (m + 1 > r and m < r + 4) else "%4.0f" % timeit2(rec, slab[m * GIB: (m + 1) * GIB].view(torch.int16), k=5))
        print("records at %3d GiB:               " % r + " ".join(row), flush=True)
    sys.exit(0)
if os.environ.get("MOVE_MASKED"):
    # records fixed at the start of the slab, the masked rows (1 GiB) moved through the rest of it
    rec = slab[:RB].view(torch.int32)
    for step in (GIB, GIB // 4, GIB // 16):
        res, off = [], RB
        while off + GIB <= slab.numel() and len(res) < 64:
            res.append(timeit2(rec, slab[off: off + GIB].view(torch.int16)))
            off += step
        print("masked rows every %d MiB from +4 GiB: " % (step >> 20) + " ".join("%.0f" % x for x in res), flush=True)
    sys.exit(0)
for step in (GIB, GIB // 4):
    res, off = [], 0
    while off + RB <= slab.numel() and len(res) < 48:
        res.append(timeit(slab[off: off + RB].view(torch.int32)))
        off += step
    print("records every %d MiB: " % (step >> 20) + " ".join("%.0f" % x for x in res), flush=True)
