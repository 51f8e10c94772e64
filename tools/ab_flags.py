#!/usr/bin/env python3
"""Same-process A/B of HRX_DEBUG_FLAGS variants on the bench line (the library reads the variable at every launch): the variants
are interleaved in blocks of launches, many times over, so that box-to-box and run-to-run drift (±5 %) cancels.
  python tools/ab_flags.py 0 32 64 96"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("HRX_LIB_PATH", os.path.join(ROOT, "halo2_regex_amd", "csrc", "libhrx_ablation.so"))   # `make -C halo2_regex_amd/csrc ablation`
sys.path.insert(0, ROOT)
import numpy as np, torch
import halo2_regex_amd as hra
from halo2_regex_amd import synth
# workload: HRX_AB_BATCH / HRX_AB_LEN / HRX_AB_ROWS (defaults: the bench line, 65536 x 1023 bytes, M = 1024)
B, N, M = int(os.environ.get("HRX_AB_BATCH", 65536)), int(os.environ.get("HRX_AB_LEN", 1023)), int(os.environ.get("HRX_AB_ROWS", 1024))
flags = sys.argv[1:] or ["0", "32", "64", "96"]
D_DIR = os.path.join(ROOT, "tests", "golden", "dfa")
rd = lambda f: open(os.path.join(D_DIR, f), "rb").read()
defs = [hra.RegexDefs(hra.AllstrRegexDef(rd("regex1_test_lookup.txt")), [hra.SubstrRegexDef(rd("substr1_test_lookup.txt"))])]
cfgs = {}
for f in flags:   # one context per variant: a release library reads HRX_DEBUG_FLAGS once, at hrx_ctx_create (kernel-selection bits only)
    os.environ["HRX_DEBUG_FLAGS"] = str(int(f, 0))
    cfgs[f] = hra.RegexVerifyConfig.configure(M, defs, device=0)
cfg = cfgs[flags[0]]
dev = torch.device("cuda", 0)
chars, lens = synth.regex1_planted(B, N, seed=0, stride=(N + 15) // 16 * 16)
d_lens = torch.from_numpy(lens.astype(np.int32)).to(dev)
d_chars = hra.chars_to_position_major(torch.from_numpy(chars).to(dev))
out = cfg.alloc_outputs_position_major(B, dev)
step = lambda: cfg.witness_batch_position_major(d_chars, d_lens, out=out, chars_pm_stride=(N + 15) // 16 * 16)

res = {f: [] for f in flags}
IT = max(3, min(60, int(60 * 65536 * 1024 / (B * M)))); WU = max(1, IT // 6)
print("workload %d x %d bytes, M = %d; %s" % (B, N, M, cfg.describe_launch(B, layout=3)))
for rep in range(12):
    for f in flags:
        os.environ["HRX_DEBUG_FLAGS"] = str(int(f, 0))
        cfg = cfgs[f]
        for _ in range(WU): step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(IT): step()
        e1.record(); torch.cuda.synchronize()
        res[f].append(e0.elapsed_time(e1) / IT * 1e3)
for f in flags:
    v = res[f][2:]
    print("flags %-10s  median %.2f us  mean %.2f  min %.2f  max %.2f" % (f, statistics.median(v), statistics.mean(v), min(v), max(v)))
