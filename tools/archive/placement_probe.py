#!/usr/bin/env python3
"""Does the bench-line launch time depend on WHERE the three buffers sit?  One process, one 3-GiB arena, the input / records /
masked buffers placed at varying offsets inside it (tools only).  Prints us per launch (60 back-to-back launches) per placement."""
import os, sys, time
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
pm = hra.chars_to_position_major(torch.from_numpy(chars).to(dev))
arena = torch.empty(3 << 30, dtype=torch.uint8, device=dev)
st = torch.empty(B, dtype=torch.int64, device=dev)
MiB = 1 << 20
def run(c_off, r_off, m_off):
    c = arena[c_off:c_off + B * 1024]; c.copy_(pm)
    r = arena[r_off:r_off + B * M * 4].view(torch.int32)
    m = arena[m_off:m_off + B * M * 2].view(torch.int16)
    step = lambda: cfg.witness_batch_position_major(c, d_lens, out=(r, m, st), chars_pm_stride=1024)
    for _ in range(10): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(60): step()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 60 * 1e3
rng = np.random.default_rng(0)
print("arena base %#x" % arena.data_ptr())
for trial in range(28):
    if trial == 0: c_off, r_off, m_off = 0, 64 * MiB, 320 * MiB                       # back to back
    elif trial < 10:                                                                   # whole-MiB shifts
        c_off, r_off, m_off = 0, (64 + int(rng.integers(0, 64))) * MiB, (400 + int(rng.integers(0, 64))) * MiB
    elif trial < 20:                                                                   # sub-MiB shifts (multiples of 4 KiB)
        c_off, r_off, m_off = int(rng.integers(0, 256)) * 4096, 128 * MiB + int(rng.integers(0, 256)) * 4096, 500 * MiB + int(rng.integers(0, 256)) * 4096
    else:                                                                              # far apart
        c_off, r_off, m_off = int(rng.integers(0, 8)) * 256 * MiB, 1024 * MiB + int(rng.integers(0, 3)) * 256 * MiB, 2048 * MiB + int(rng.integers(0, 3)) * 256 * MiB
        c_off += 2 * 1024 * MiB if False else 0
    t = run(c_off, r_off, m_off)
    print("chars +%-12d records +%-12d masked +%-12d  %.1f us" % (c_off, r_off, m_off, t))
t = [run(0, 64 * MiB, 320 * MiB) for _ in range(3)]
print("back to back again:", " ".join("%.1f" % x for x in t))
