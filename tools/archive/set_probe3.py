#!/usr/bin/env python3
"""Launch time by how the buffers were ALLOCATED (tools only): torch (hipMalloc) against hrx_device_alloc with different chunk
sizes / orders (HRX_ALLOC is honoured by the prototype allocator).  usage: set_probe3.py [regex23|regex1|regex1x4]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import halo2_regex_amd as hra
from halo2_regex_amd import synth
which = sys.argv[1] if len(sys.argv) > 1 else "regex23"
D_DIR = os.path.join(ROOT, "tests", "golden", "dfa")
rd = lambda f: open(os.path.join(D_DIR, f), "rb").read()
mk = lambda k: hra.RegexDefs(hra.AllstrRegexDef(rd("regex%d_test_lookup.txt" % k)), [hra.SubstrRegexDef(rd("substr%d_test_lookup.txt" % k))])
if which == "regex23": B, N, M, defs, gen, K = 262144, 2047, 2048, [mk(2), mk(3)], synth.regex23_planted, 20
elif which == "regex1": B, N, M, defs, gen, K = 65536, 1023, 1024, [mk(1)], synth.regex1_planted, 100
else: B, N, M, defs, gen, K = 262144, 1023, 1024, [mk(1)], synth.regex1_planted, 50
cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
dev = torch.device("cuda", 0)
chars, lens = gen(B, N, seed=0, stride=M)
d_lens = torch.from_numpy(lens.astype(np.int32)).to(dev)
pm0 = hra.chars_to_position_major(torch.from_numpy(chars).to(dev))
r0, m0, st = cfg.alloc_outputs_position_major(B, dev)
def timeit(pm, out, k=K):
    for _ in range(3): cfg.witness_batch_position_major(pm, d_lens, out=out, chars_pm_stride=M)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): cfg.witness_batch_position_major(pm, d_lens, out=out, chars_pm_stride=M)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
def rotating(sets, k=K):
    for pm, out in sets: cfg.witness_batch_position_major(pm, d_lens, out=out, chars_pm_stride=M)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(k):
        pm, out = sets[i % len(sets)]
        cfg.witness_batch_position_major(pm, d_lens, out=out, chars_pm_stride=M)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
NS = int(os.environ.get("PROBE_NS", "3" if which != "regex1" else "8"))
for mode in os.environ.get("PROBE_MODES", "torch plain chunk:16 chunk:2 chunk:4 chunk:64 order:16 order:2 chunk:16").split():
    os.environ["HRX_ALLOC"] = mode
    sets = []
    for s in range(NS):
        if mode == "torch":
            rec, msk = torch.empty_like(r0), torch.empty_like(m0); pm = pm0.clone()
        else:
            rec, msk = hra.device_empty(r0.numel(), torch.int32, dev), hra.device_empty(m0.numel(), torch.int16, dev)
            pm = hra.device_empty(pm0.numel(), torch.uint8, dev); pm.copy_(pm0)
        sets.append((pm, (rec, msk, st)))
    print("%-9s outputs+input: %s | outputs only: %s | rotating over the sets: %.1f us" % (
        mode, " ".join("%7.1f" % timeit(pm, out) for pm, out in sets), " ".join("%7.1f" % timeit(pm0, out) for _, out in sets), rotating(sets)), flush=True)
    if os.environ.get("PROBE_PTRS"): print("          " + "  ".join("rec %#x msk %#x in %#x" % (out[0].data_ptr(), out[1].data_ptr(), pm.data_ptr()) for pm, out in sets))
    del sets
