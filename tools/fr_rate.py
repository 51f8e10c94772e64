#!/usr/bin/env python3
"""SURVEY §8 f4 measurement: rate of hrx_fr_columns_device (compact witness -> bn256::Fr Montgomery cells) on one MI355X.
Write-bound: 32 B per cell x (4 + 4 D) cells per row; reads are the 7 B/row compact witness."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import halo2_regex_amd as hra
from halo2_regex_amd import synth
D_DIR = os.path.join(ROOT, "tests", "golden", "dfa")
rd = lambda f: open(os.path.join(D_DIR, f), "rb").read()
B, n, M, NB = 65536, 1023, 1024, 8192      # witness for 65536 strings, cells for 8192 of them per call (2.1 GB of cells)
defs = [hra.RegexDefs(hra.AllstrRegexDef(rd("regex1_test_lookup.txt")), [hra.SubstrRegexDef(rd("substr1_test_lookup.txt"))])]
cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
dev = torch.device("cuda", 0)
chars, lens = synth.regex1_planted(B, n, seed=0, stride=1024)
SM = len(sys.argv) > 1 and sys.argv[1] == "sm"      # the compact witness in string-major buffers instead of position-major ones
d_chars = torch.from_numpy(chars).to(dev)
d_lens = torch.from_numpy(lens.astype(np.int32)).to(dev)
if SM:
    out = cfg.witness_batch(d_chars, d_lens)
    kw = dict()
else:
    d_chars = hra.chars_to_position_major(d_chars)
    out = cfg.witness_batch_position_major(d_chars, d_lens, chars_pm_stride=1024)
    kw = dict(position_major=True, chars_pm_stride=1024)
for _ in range(2):
    cells = cfg.fr_columns(d_chars, d_lens, out, b_begin=0, b_count=NB, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
steps = 10
e0.record()
for i in range(steps):
    cells = cfg.fr_columns(d_chars, d_lens, out, b_begin=(i * NB) % B, b_count=NB, **kw)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / steps
rows = NB * M
cells_per_row = 8
print(json.dumps({"what": "hrx_fr_columns_device, regex1 (D=1), %d strings x %d rows per call, %s witness" % (NB, M, "string-major" if SM else "position-major"), "ms_per_call": ms,
                  "rows_per_s": rows / (ms * 1e-3), "cells_per_s": rows * cells_per_row / (ms * 1e-3),
                  "written_GBps": rows * cells_per_row * 32 / (ms * 1e-3) / 1e9, "frac_of_8TBps": rows * (cells_per_row * 32 + 7) / (ms * 1e-3) / 8e12}))
