#!/usr/bin/env python3
"""In-kernel stamps of the bench-line launch in the ROTATING-buffer regime (profiling only; needs `make -C halo2_regex_amd/csrc stamps`):
  HRX_LIB_PATH=halo2_regex_amd/csrc/libhrx_stamps.so python3 tools/stamps_rotating.py [nsets] [launches]
Runs `launches` back-to-back launches over `nsets` buffer sets and prints, for the LAST launch, when its walkers entered / started walking /
finished (100-MHz wall clock, us from the first workgroup's entry), per XCD, and where the walkers' cycles went."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import halo2_regex_amd as hra
from halo2_regex_amd import synth
nsets = int(sys.argv[1]) if len(sys.argv) > 1 else 8
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 40
assert hasattr(hra.lib, "hrx_debug_read_stamps"), "load libhrx_stamps.so through HRX_LIB_PATH"
dev = torch.device("cuda", 0)
D = os.path.join(ROOT, "tests", "golden", "dfa")
rd = lambda f: open(os.path.join(D, f), "rb").read()
cfg = hra.RegexVerifyConfig.configure(1024, [hra.RegexDefs(hra.AllstrRegexDef(rd("regex1_test_lookup.txt")), [hra.SubstrRegexDef(rd("substr1_test_lookup.txt"))])], device=0)
B, M, stride = 65536, 1024, 1024
chars, lens = synth.noise(B, 1023, seed=0, stride=stride)
d_l = torch.from_numpy(lens.astype(np.int32)).to(dev)
d_c0 = torch.from_numpy(chars).to(dev)
sets = []
for k in range(nsets):
    sets.append((hra.chars_to_position_major(torch.roll(d_c0, k * 8229, 0)), cfg.alloc_outputs_position_major(B, dev)))
run = lambda i: cfg.witness_batch_position_major(sets[i % nsets][0], d_l, out=sets[i % nsets][1], chars_pm_stride=stride)
for i in range(2 * nsets):
    run(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(launches):
    run(i)
e1.record()
torch.cuda.synchronize()
print("%d launches over %d sets: %.2f us per launch (events; eager launches, stamps build)" % (launches, nsets, e0.elapsed_time(e1) * 1e3 / launches))
npairs = 1024
buf = (C.c_ulonglong * (npairs * 16))()
hra.lib.hrx_debug_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_size_t]
assert hra.lib.hrx_debug_read_stamps(cfg._ctx, buf, npairs * 16) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(npairs, 16).astype(np.int64)
wait, walk, end, grp, w_entry, w_start, w_end = (s[:, i] for i in range(7))
t0 = w_entry.min()
us = lambda x: (x - t0) * 0.01
pct = lambda x: " ".join("%.1f" % v for v in np.percentile(x, [0, 10, 50, 90, 100]))
print("entry  (us, pct 0/10/50/90/100): " + pct(us(w_entry)))
print("start walking                  : " + pct(us(w_start)))
print("walker done                    : " + pct(us(w_end)))
wg = np.arange(npairs) // 4
for x in range(8):
    m = (wg % 8) == x
    print("  XCD %d: start %.1f  done median %.1f  max %.1f   walk-cycles %.0fk wait %.0fk end %.0fk" % (x, np.median(us(w_start)[m]), np.median(us(w_end)[m]), us(w_end)[m].max(),
          walk[m].mean() / 1e3, wait[m].mean() / 1e3, end[m].mean() / 1e3))
print("walker cycles per group: wait %.0fk walk %.0fk tile-end %.0fk total %.0fk" % (wait.mean() / 1e3, walk.mean() / 1e3, end.mean() / 1e3, grp.mean() / 1e3))
