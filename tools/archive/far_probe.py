#!/usr/bin/env python3
"""cfg 3 shape: masked rows allocated right after the records (what a fresh process gets) against masked rows allocated while a
spacer holds all the free device memory but theirs — i.e. at the other end of the device memory.  (tools only)
usage: far_probe.py [batch = 262144]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import halo2_regex_amd as hra
from halo2_regex_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
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
def timeit(msk, k=8):
    for _ in range(2): cfg.witness_batch_position_major(pm0, d_lens, out=(rec, msk, st), chars_pm_stride=2048)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): cfg.witness_batch_position_major(pm0, d_lens, out=(rec, msk, st), chars_pm_stride=2048)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
near = torch.empty(B * M, dtype=torch.int16, device=dev)
res = ["near %.0f" % timeit(near)]
for frac in (0.25, 0.5, 0.75, 1.0):
    free, total = torch.cuda.mem_get_info()
    t0 = time.time()
    room = free - B * M * 2 - (2 << 30)
    spacer = torch.empty(int(room * frac), dtype=torch.uint8, device=dev)
    far = torch.empty(B * M, dtype=torch.int16, device=dev)
    del spacer
    torch.cuda.empty_cache(); torch.cuda.synchronize()
    dt = time.time() - t0
    res.append("spacer %.0f GiB (%.2f s): %.0f" % (room * frac / 2**30, dt, timeit(far)))
    del far; torch.cuda.empty_cache()
print("B=%d, us per launch: " % B + " | ".join(res))
