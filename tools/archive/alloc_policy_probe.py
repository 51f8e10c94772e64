#!/usr/bin/env python3
"""Launch time by the policy of hrx_device_alloc (HRX_ALLOC_POSITIONS x HRX_ALLOC_SPACER_MIB x HRX_ALLOC_CHUNK_MIB are
honoured by the allocator for this probe) against torch's hipMalloc buffers.  (tools only)
usage: alloc_policy_probe.py [regex23 | regex1 | regex1x4 | headers3]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import halo2_regex_amd as hra
from halo2_regex_amd import synth
which = sys.argv[1] if len(sys.argv) > 1 else "regex23"
D_DIR = os.path.join(ROOT, "tests", "golden", "dfa")
rd = lambda f: open(os.path.join(D_DIR, f), "rb").read()
mk = lambda k: hra.RegexDefs(hra.AllstrRegexDef(rd("regex%d_test_lookup.txt" % k)), [hra.SubstrRegexDef(rd("substr%d_test_lookup.txt" % k))])
if which == "regex23": B, N, M, defs, gen, K, NS = 262144, 2047, 2048, [mk(2), mk(3)], synth.regex23_planted, 12, 1
elif which == "regex1": B, N, M, defs, gen, K, NS = 65536, 1023, 1024, [mk(1)], synth.regex1_planted, 160, 8
else: B, N, M, defs, gen, K, NS = 262144, 1023, 1024, [mk(1)], synth.regex1_planted, 40, 1
cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
dev = torch.device("cuda", 0)
chars, lens = gen(B, N, seed=0, stride=M)
d_lens = torch.from_numpy(lens.astype(np.int32)).to(dev)
pm0 = hra.chars_to_position_major(torch.from_numpy(chars).to(dev))
r0, m0, st = cfg.alloc_outputs_position_major(B, dev)
NR, NM = r0.numel(), m0.numel()
del r0, m0
def run(sets, k=K):
    for pm, out in sets: cfg.witness_batch_position_major(pm, d_lens, out=out, chars_pm_stride=M)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(k):
        pm, out = sets[i % len(sets)]
        cfg.witness_batch_position_major(pm, d_lens, out=out, chars_pm_stride=M)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
alg = float(lens.sum()) * (1 + 4 * len(defs) + 2)
for pol in os.environ.get("POLICIES", "torch 3x98304x2 8x32768x2 16x16384x2 32x8192x2 64x4096x2 16x16384x16 16x1024x2 16x0x2").split():
    t0 = time.time()
    sets = []
    for s in range(NS):
        if pol == "torch":
            rec, msk, pm = torch.empty(NR, dtype=torch.int32, device=dev), torch.empty(NM, dtype=torch.int16, device=dev), pm0.clone()
        else:
            os.environ["HRX_ALLOC_POSITIONS"], os.environ["HRX_ALLOC_SPACER_MIB"], os.environ["HRX_ALLOC_CHUNK_MIB"] = pol.split("x")
            rec, msk = hra.device_empty(NR, torch.int32, dev, chunked=True), hra.device_empty(NM, torch.int16, dev, chunked=True)
            pm = hra.device_empty(pm0.numel(), torch.uint8, dev, chunked=True); pm.copy_(pm0)
        sets.append((pm, (rec, msk, st)))
    torch.cuda.synchronize(); ta = time.time() - t0
    t = [run(sets), run(sets)]
    print("%-14s (positions x spacer MiB x chunk MiB)  %7.1f %7.1f us  = %.3f of 8 TB/s   [%d set(s) in turn, allocated in %.2f s]" % (pol, t[0], t[1], alg / min(t) / 8e6, NS, ta), flush=True)
    del sets, rec, msk, pm
