#!/usr/bin/env python3
"""Same-process A/B of the def-parallel kernel's write-front gate (HRX_GATE_W, read at context creation): one context per slack value over the SAME buffer sets,
timing blocks alternating.  python3 tools/gate_ab.py [B] [M] [W ...]   (W = 0: no gate)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import halo2_regex_amd as hra
from halo2_regex_amd import synth
import bench
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
M = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
Ws = [int(x) for x in sys.argv[3:]] or [0, 8, 16, 24, 48, 96]
cfgname = os.environ.get("GATE_CONFIG", "headers3")
names = bench.workload(bench.parse_args(["--config", cfgname]))[0]
D = len(names)
defs = lambda: [hra.RegexDefs(hra.AllstrRegexDef(a), [hra.SubstrRegexDef(t) for t in subs]) for a, subs in names]
cfgs = {}
for w in Ws:
    os.environ["HRX_GATE_W"] = str(abs(w) % 1000 if w != -1 else 0)     # negative: the round-4 dealing of the wave roles (-1: no gate)
    cfgs[w] = hra.RegexVerifyConfig.configure(M, defs(), device=0)
print("%s %d x %d: %s" % (cfgname, B, M, cfgs[Ws[0]].describe_launch(B, layout=3)))
nd = min(B, 4096)
gen = synth.headers_planted if cfgname.startswith("headers") else synth.regex23_planted
base_c, base_l = gen(nd, M - 1, seed=3, stride=M)
d_c = torch.from_numpy(base_c).to(dev)
d_l0 = torch.from_numpy(base_l.astype(np.int32)).to(dev)
d_c = torch.cat([torch.roll(d_c, shifts=131 * j, dims=0) for j in range(B // nd)])
d_l = torch.cat([torch.roll(d_l0, shifts=131 * j, dims=0) for j in range(B // nd)])
d_c = hra.chars_to_position_major(d_c)
nsets = 3
outs = [cfgs[Ws[0]].alloc_outputs_position_major(B, dev) for _ in range(nsets)]
print("placement best GB/s:", [round(cfgs[Ws[0]].last_placement_report()["best_gbs"])])
K = 6
res = {w: [] for w in Ws}
ref = None
for rnd in range(5):
    for w in Ws:
        c = cfgs[w]
        if w < 0: os.environ["HRX_PMD_ROLES_OLD"] = "1"
        else: os.environ.pop("HRX_PMD_ROLES_OLD", None)
        os.environ["HRX_PMD_PRIO"] = str((abs(w) // 1000) % 10)      # thousands digit: priority experiment mode
        for i in range(nsets):
            c.witness_batch_position_major(d_c, d_l, out=outs[i], chars_pm_stride=M)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(K):
            c.witness_batch_position_major(d_c, d_l, out=outs[i % nsets], chars_pm_stride=M)
        e1.record(); torch.cuda.synchronize()
        res[w].append(e0.elapsed_time(e1) / K)
        if rnd == 0:
            cur = [o.clone() for o in outs[0]]
            if ref is None: ref = cur
            else: assert all(torch.equal(x, y) for x, y in zip(ref, cur)), "outputs differ with gate W=%d" % w
            del cur
rows = int(d_l.sum())
for w in Ws:
    v = sorted(res[w][1:]); med = v[len(v) // 2]
    print("  W=%-4d ms/launch %s  median %.4f  frac %.3f" % (w, " ".join("%.4f" % x for x in res[w]), med, rows * (4 * D + 3) / (med * 1e-3) / 8e12))
