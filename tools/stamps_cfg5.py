#!/usr/bin/env python3
"""In-kernel stamps of the cfg 5 launch (256-state DFA, HALF table; profiling only; `make -C halo2_regex_amd/csrc stamps`):
  HRX_LIB_PATH=halo2_regex_amd/csrc/libhrx_stamps.so python3 tools/stamps_cfg5.py [B] [n]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import halo2_regex_amd as hra
from halo2_regex_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
assert hasattr(hra.lib, "hrx_debug_read_stamps"), "load libhrx_stamps.so through HRX_LIB_PATH"
dev = torch.device("cuda", 0)
allb = np.arange(256, dtype=np.uint8)
a_txt, sub_txt = synth.random_dfa(256, seed=2, alphabet=allb, n_substr_pairs=200)
cfg = hra.RegexVerifyConfig.configure(n, [hra.RegexDefs(hra.AllstrRegexDef(a_txt), [hra.SubstrRegexDef(sub_txt)])], device=0)
print(cfg.describe_launch(B, layout=3))
chars, lens = synth.noise(B, n, seed=0, alphabet=allb, stride=n)
d_l = torch.from_numpy(lens.astype(np.int32)).to(dev)
d_c = hra.chars_to_position_major(torch.from_numpy(chars).to(dev))
out = cfg.alloc_outputs_position_major(B, dev)
run = lambda: cfg.witness_batch_position_major(d_c, d_l, out=out, chars_pm_stride=n)
for i in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(10):
    run()
e1.record()
torch.cuda.synchronize()
print("%.1f us per launch (events; eager launches, stamps build)" % (e0.elapsed_time(e1) * 1e3 / 10))
npairs = 1024
buf = (C.c_ulonglong * (npairs * 16))()
hra.lib.hrx_debug_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_size_t]
assert hra.lib.hrx_debug_read_stamps(cfg._ctx, buf, npairs * 16) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(npairs, 16).astype(np.int64)
wait, walk, end, grp, w_entry, w_start, w_end = (s[:, i] for i in range(7))
t0 = w_entry.min()
us = lambda x: (x - t0) * 0.01
pct = lambda x: " ".join("%.1f" % v for v in np.percentile(x, [0, 10, 50, 90, 100]))
print("start walking (us): " + pct(us(w_start)) + "   walker done: " + pct(us(w_end)))
nt = (n + 63) // 64
print("walker cycles per TILE: input wait %.0f  walk (incl. record stores) %.0f  tile end (masks, repairs, masked rows) %.0f  total %.0f" %
      (wait.mean() / nt, walk.mean() / nt, end.mean() / nt, grp.mean() / nt))
print("finisher cycles per TILE: waiting for the walker's summary %.0f  work (masks, repairs, masked rows) %.0f" % (s[:, 8].mean() / nt, s[:, 9].mean() / nt))
print("  of the work: summary reads + bitvectors %.0f  mask scans %.0f  held-row edits + repairs at the memory %.0f  masked rows (assembly, hold, stores) %.0f" %
      tuple(s[:, 10 + i].mean() / nt for i in range(4)))
