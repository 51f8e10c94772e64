#!/usr/bin/env python3
"""Time series of the bench-line kernel under sustained load: a hipGraph of CH launches replayed R times back to back, one
event pair per replay (tools only).  Shows whether the launch time holds once the chip has been busy for tens of ms.
  python3 tools/sustained.py [R] [CH] [gap_ms]      gap_ms > 0: host sleep between replays (a duty cycle below 100 %)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import halo2_regex_amd as hra
from halo2_regex_amd import synth
R, CH, GAP = int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 40, float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
B, N, M = int(os.environ.get("HRX_AB_BATCH", 65536)), int(os.environ.get("HRX_AB_LEN", 1023)), int(os.environ.get("HRX_AB_ROWS", 1024))
D_DIR = os.path.join(ROOT, "tests", "golden", "dfa")
rd = lambda f: open(os.path.join(D_DIR, f), "rb").read()
defs = [hra.RegexDefs(hra.AllstrRegexDef(rd("regex1_test_lookup.txt")), [hra.SubstrRegexDef(rd("substr1_test_lookup.txt"))])]
cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
dev = torch.device("cuda", 0)
stride = (N + 15) // 16 * 16
chars, lens = synth.regex1_planted(B, N, seed=0, stride=stride)
d_lens = torch.from_numpy(lens.astype(np.int32)).to(dev)
d_chars = hra.chars_to_position_major(torch.from_numpy(chars).to(dev))
out = cfg.alloc_outputs_position_major(B, dev)
step = lambda: cfg.witness_batch_position_major(d_chars, d_lens, out=out, chars_pm_stride=stride)
for _ in range(5): step()
torch.cuda.synchronize()
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        for _ in range(CH): step()
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize()
time.sleep(0.5)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * R)]
for i in range(R):
    ev[2 * i].record(); g.replay(); ev[2 * i + 1].record()
    if GAP > 0:
        torch.cuda.synchronize(); time.sleep(GAP * 1e-3)
torch.cuda.synchronize()
print("%s; graph of %d launches x %d replays, gap %.1f ms: us/launch per replay:" % (cfg.describe_launch(B, layout=3), CH, R, GAP),
      " ".join("%.1f" % (ev[2 * i].elapsed_time(ev[2 * i + 1]) / CH * 1e3) for i in range(R)))
