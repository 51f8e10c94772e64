#!/usr/bin/env python3
"""cfg 3 shape: records / masked rows / input carved out of ONE device allocation at chosen relative offsets (tools only).
set_probe.py showed the launch time (0.97 .. 1.19 ms) going with the OUTPUT buffers, not with the process or the input."""
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
r0, m0, st = cfg.alloc_outputs_position_major(B, dev)
RB, MB, CB = r0.numel() * 4, m0.numel() * 2, pm0.numel()
PAD = 256 << 20
def timeit(pm, out, k=20):
    for _ in range(3): cfg.witness_batch_position_major(pm, d_lens, out=out, chars_pm_stride=2048)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): cfg.witness_batch_position_major(pm, d_lens, out=out, chars_pm_stride=2048)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
print("separate allocations: %.3f" % timeit(pm0, (r0, m0, st)))
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    slab = torch.empty(RB + MB + CB + 3 * PAD, dtype=torch.uint8, device=dev)
    base = slab.data_ptr()
    al = (-base) % (1 << 21)      # start at a 2-MiB boundary of the allocation
    def carve(off, nbytes, dtype): return slab[al + off: al + off + nbytes].view(dtype)
    res = []
    for d_m in (0, 4 << 10, 64 << 10, 1 << 20, (1 << 20) + (4 << 10), 3 << 20, 17 << 20, 64 << 20, 129 << 20):
        rec = carve(0, RB, torch.int32); msk = carve(RB + d_m, MB, torch.int16)
        res.append("%s:%.3f" % (("%dK" % (d_m >> 10)), timeit(pm0, (rec, msk, st))))
    print("slab %#x  masked at records_end + d: " % base + "  ".join(res))
    res = []
    for d_r in (0, 4 << 10, 1 << 20, 5 << 20, 64 << 20):   # records shifted inside the slab, masked fixed behind
        rec = carve(d_r, RB, torch.int32); msk = carve(RB + PAD, MB, torch.int16)
        res.append("%dK:%.3f" % (d_r >> 10, timeit(pm0, (rec, msk, st))))
    print("   records at d, masked fixed: " + "  ".join(res))
    pmc = carve(RB + PAD + MB + PAD, CB, torch.uint8); pmc.copy_(pm0)
    rec = carve(0, RB, torch.int32); msk = carve(RB, MB, torch.int16)
    print("   input in the slab too: %.3f" % timeit(pmc, (rec, msk, st)))
    keep = slab if trial == 0 else None
