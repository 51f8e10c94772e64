#!/usr/bin/env python3
"""One fresh process = one sample of the bench-line launch time and of where the buffers landed (tools only): the time has a
per-process state (70 vs 78 us); does it go with the addresses?"""
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
pre = int(sys.argv[1]) if len(sys.argv) > 1 else 0
junk = [torch.empty(pre << 20, dtype=torch.uint8, device=dev)] if pre else []     # shift what the allocator hands out next
chars, lens = synth.regex1_planted(B, N, seed=0, stride=1024)
d_lens = torch.from_numpy(lens.astype(np.int32)).to(dev)
pm = hra.chars_to_position_major(torch.from_numpy(chars).to(dev))
out = cfg.alloc_outputs_position_major(B, dev)
step = lambda: cfg.witness_batch_position_major(pm, d_lens, out=out, chars_pm_stride=1024)
for _ in range(20): step()
torch.cuda.synchronize()
res = []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): step()
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) * 10)
print("us/launch %s  chars %#x rec %#x msk %#x  pre %d MiB" % (" ".join("%.1f" % x for x in res), pm.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), pre))
