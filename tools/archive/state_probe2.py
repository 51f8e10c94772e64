#!/usr/bin/env python3
"""Several independently allocated buffer sets in ONE process: is the 70-vs-78-us state a property of the process or of the
buffers' physical pages?  (tools only)"""
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
sets = []
for k in range(6):
    pm = pm0.clone()
    sets.append((pm, cfg.alloc_outputs_position_major(B, dev)))
def t(pm, out):
    step = lambda: cfg.witness_batch_position_major(pm, d_lens, out=out, chars_pm_stride=1024)
    for _ in range(20): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): step()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 10
for rnd in range(2):
    print("round %d:" % rnd, "  ".join("set%d %.1f" % (k, t(*s)) for k, s in enumerate(sets)))
# mixed: input of set i with outputs of set j
print("mixed:", "  ".join("in%d/out%d %.1f" % (i, j, t(sets[i][0], sets[j][1])) for i, j in ((0, 1), (1, 0), (2, 3), (3, 2), (4, 5), (5, 4))))
