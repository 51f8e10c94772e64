#!/usr/bin/env python3
"""Speed map of one big allocation (tools only): the same one-block launch (regex2+regex3, 65536 strings x M rows) with its
records at consecutive offsets of a 32-GiB slab.  set_probe4.py: the fast / slow state of the cfg 3 launch belongs to the
RECORDS buffer and comes in levels, as if every GiB of it were either fast or slow."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import halo2_regex_amd as hra
from halo2_regex_amd import synth
B = 65536
D_DIR = os.path.join(ROOT, "tests", "golden", "dfa")
rd = lambda f: open(os.path.join(D_DIR, f), "rb").read()
defs = [hra.RegexDefs(hra.AllstrRegexDef(rd("regex%d_test_lookup.txt" % k)), [hra.SubstrRegexDef(rd("substr%d_test_lookup.txt" % k))]) for k in (2, 3)]
dev = torch.device("cuda", 0)
GIB = 1 << 30
slab = torch.empty(int(os.environ.get("SLAB_GIB", "32")) * GIB + (4 << 20), dtype=torch.uint8, device=dev)
al = (-slab.data_ptr()) % (2 << 20)
print("slab %#x (+%d to the 2-MiB boundary)" % (slab.data_ptr(), al))
for M, step in ((2048, GIB), (2048, GIB // 2), (1024, GIB // 2), (512, GIB // 4)):
    cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
    chars, lens = synth.regex23_planted(B, M - 1, seed=0, stride=M)
    d_lens = torch.from_numpy(lens.astype(np.int32)).to(dev)
    pm0 = hra.chars_to_position_major(torch.from_numpy(chars).to(dev))
    r0, m0, st = cfg.alloc_outputs_position_major(B, dev)
    RB = r0.numel() * 4
    def timeit(rec, k=16):
        for _ in range(2): cfg.witness_batch_position_major(pm0, d_lens, out=(rec, m0, st), chars_pm_stride=M)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k): cfg.witness_batch_position_major(pm0, d_lens, out=(rec, m0, st), chars_pm_stride=M)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / k * 1e3
    res = []
    off = 0
    while off + RB <= slab.numel() - al - (2 << 20):
        res.append(timeit(slab[al + off: al + off + RB].view(torch.int32)))
        off += step
    print("M=%d (records %d MiB) every %d MiB: " % (M, RB >> 20, step >> 20) + " ".join("%.0f" % x for x in res), flush=True)
