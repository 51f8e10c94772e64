#!/usr/bin/env python3
"""Configs of 4 .. 7 defs, position-major: ONE def-parallel launch over the CLASS-WIDE tables (hrx_kernel_pmd.hip CW) against the passes over groups of three defs (HRX_DEBUG_FLAGS=0x2000000:
kDbgNoDefParallel), same process, same buffers, blocks alternating; the two paths' outputs compared bit for bit.  python3 tools/dn_bench.py [B] [M]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import halo2_regex_amd as hra
from halo2_regex_amd import synth
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
M = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
DFA = os.path.join(ROOT, "tests", "golden", "dfa")
rd = lambda f: open(os.path.join(DFA, f), "rb").read()
pair = lambda k: (rd("regex%d_test_lookup.txt" % k), [rd("substr%d_test_lookup.txt" % k)])
hdr = lambda n, ns: (rd(n + "_lookup.txt"), [rd("%s_substr%d.txt" % (n, k)) for k in range(ns)])
H = [hdr("header_from", 1), hdr("header_to", 1), hdr("header_subject", 3)]
cfgs = {4: H + [pair(1)], 5: H + [pair(1), pair(2)], 6: H + [pair(1), pair(2), pair(3)], 7: H + [pair(1), pair(2), pair(3), hdr("header_from", 1)],
        8: H + [pair(1), pair(2), pair(3), hdr("header_from", 1), hdr("header_to", 1)]}
base_c, base_l = synth.headers_planted(4096, M - 1, seed=3, stride=M)
d_c = torch.from_numpy(base_c).to(dev)
d_c = torch.cat([torch.roll(d_c, shifts=131 * j, dims=0) for j in range(B // 4096)])
d_l0 = torch.from_numpy(base_l.astype(np.int32)).to(dev)
d_l = torch.cat([torch.roll(d_l0, shifts=131 * j, dims=0) for j in range(B // 4096)])
d_c = hra.chars_to_position_major(d_c)
rows = int(d_l.sum())
for D, names in cfgs.items():
    mk = lambda: hra.RegexVerifyConfig.configure(M, [hra.RegexDefs(hra.AllstrRegexDef(a), [hra.SubstrRegexDef(t) for t in subs]) for a, subs in names], device=0)
    os.environ.pop("HRX_DEBUG_FLAGS", None); one = mk()
    os.environ["HRX_DEBUG_FLAGS"] = "0x2000000"; passes = mk()
    os.environ.pop("HRX_DEBUG_FLAGS", None)
    nsets = 4
    outs = [one.alloc_outputs_position_major(B, dev) for _ in range(nsets)]
    ref = passes.alloc_outputs_position_major(B, dev)
    passes.witness_batch_position_major(d_c, d_l, out=ref, chars_pm_stride=M)
    one.witness_batch_position_major(d_c, d_l, out=outs[0], chars_pm_stride=M)
    torch.cuda.synchronize()
    same = all(torch.equal(x, y) for x, y in zip(outs[0], ref))
    res = {}
    for name, c in (("one launch", one), ("passes", passes)) * 3:
        for i in range(nsets): c.witness_batch_position_major(d_c, d_l, out=outs[i], chars_pm_stride=M)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        K = 12
        e0.record()
        for i in range(K): c.witness_batch_position_major(d_c, d_l, out=outs[i % nsets], chars_pm_stride=M)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(name, []).append(e0.elapsed_time(e1) / K)
    bpr = 4 * D + 3
    print("D=%d %d x %d  equal outputs: %s   %s" % (D, B, M, same, one.describe_launch(B, layout=3)[:70]))
    for name in ("one launch", "passes"):
        m = sorted(res[name])[1]
        print("   %-10s %s  median %.4f ms  frac %.3f" % (name, " ".join("%.4f" % x for x in res[name]), m, rows * bpr / (m * 1e-3) / 8e12))
    del outs, ref, one, passes
