#!/usr/bin/env python3
"""The bench-line kernel writing into ROTATING output buffers (tools only): with one output set re-written by every launch, lines
stored write-back can be overwritten in the 256-MB Infinity Cache before they ever reach HBM; with NSETS sets of 384 MiB used in
turn they cannot.  us per launch for HRX_NT_MIX (ablation build) given in the environment."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("HRX_LIB_PATH", os.path.join(ROOT, "halo2_regex_amd", "csrc", "libhrx_ablation.so"))
sys.path.insert(0, ROOT)
import numpy as np, torch
import halo2_regex_amd as hra
from halo2_regex_amd import synth
NSETS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B, N, M = 65536, 1023, 1024
D_DIR = os.path.join(ROOT, "tests", "golden", "dfa")
rd = lambda f: open(os.path.join(D_DIR, f), "rb").read()
defs = [hra.RegexDefs(hra.AllstrRegexDef(rd("regex1_test_lookup.txt")), [hra.SubstrRegexDef(rd("substr1_test_lookup.txt"))])]
cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
dev = torch.device("cuda", 0)
ins, outs = [], []
for k in range(NSETS):
    chars, lens = synth.regex1_planted(B, N, seed=k, stride=1024)
    ins.append((hra.chars_to_position_major(torch.from_numpy(chars).to(dev)), torch.from_numpy(lens.astype(np.int32)).to(dev)))
    outs.append(cfg.alloc_outputs_position_major(B, dev))
def step(i):
    c, l = ins[i % NSETS]
    cfg.witness_batch_position_major(c, l, out=outs[i % NSETS], chars_pm_stride=1024)
for i in range(2 * NSETS): step(i)
torch.cuda.synchronize()
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
g = torch.cuda.CUDAGraph()
K = 50 * NSETS
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        for i in range(K): step(i)
torch.cuda.current_stream(dev).wait_stream(side)
g.replay(); torch.cuda.synchronize()
res = []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / K * 1e3)
print("HRX_NT_MIX=%s, %d input / output sets used in turn: %s us per launch" % (os.environ.get("HRX_NT_MIX", "(default)"), NSETS, " ".join("%.1f" % x for x in res)))
