#!/usr/bin/env python3
"""dynamic group assignment under HIP-graph replay with SEVERAL output buffer sets: which launches of the replay did their work?"""
import os, sys
os.environ["HRX_DEBUG_FLAGS"] = str(0x1000 | 0x20000000)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import halo2_regex_amd as hra
from halo2_regex_amd import synth
from oracle_lib import DFA_DIR
CFG = [["regex1_test_lookup.txt", ["substr1_test_lookup.txt"]]]
defs = [hra.RegexDefs(hra.AllstrRegexDef.read_from_text(os.path.join(DFA_DIR, a)), [hra.SubstrRegexDef.read_from_text(os.path.join(DFA_DIR, s)) for s in subs]) for a, subs in CFG]
M, B, NS = 256, 200000, 4
cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
print(cfg.describe_launch(B, layout=3))
dev = torch.device("cuda", 0)
chars, lens = synth.noise(B, M - 1, seed=1, stride=M)
d_c = hra.chars_to_position_major(torch.from_numpy(chars).to(dev)); d_l = torch.from_numpy(lens.astype(np.int32)).to(dev)
outs = [cfg.alloc_outputs_position_major(B, dev) for _ in range(NS)]
for o in outs: cfg.witness_batch_position_major(d_c, d_l, out=o, chars_pm_stride=M)
torch.cuda.synchronize()
side = torch.cuda.Stream(device=dev); side.wait_stream(torch.cuda.current_stream(dev))
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        for o in outs: cfg.witness_batch_position_major(d_c, d_l, out=o, chars_pm_stride=M)
torch.cuda.current_stream(dev).wait_stream(side)
for rep in range(3):
    for o in outs: o[2].fill_(-1)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    print("replay %d: strings left unwritten per set:" % rep, [int((o[2] == -1).sum()) for o in outs])
for o in outs: o[2].fill_(-1)
torch.cuda.synchronize()
for o in outs: cfg.witness_batch_position_major(d_c, d_l, out=o, chars_pm_stride=M)
torch.cuda.synchronize()
print("eager: strings left unwritten per set:", [int((o[2] == -1).sum()) for o in outs])
