#!/usr/bin/env python3
"""The 70-vs-78-us state of the bench-line kernel per HIP stream of one process (tools only): if it follows the stream, it is a
property of the hardware queue (workgroup -> XCD rotation), not of buffers or process."""
import os, sys
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
out = cfg.alloc_outputs_position_major(B, dev)
torch.cuda.synchronize()
streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in range(11)]
for rnd in range(2):
    line = []
    for k, s in enumerate(streams):
        with torch.cuda.stream(s):
            step = lambda: cfg.witness_batch_position_major(pm, d_lens, out=out, chars_pm_stride=1024, stream=s)
            for _ in range(20): step()
            s.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(100): step()
            e1.record(s); s.synchronize()
            line.append("%.1f" % (e0.elapsed_time(e1) * 10))
    print("round %d, us/launch per stream:" % rnd, " ".join(line))
