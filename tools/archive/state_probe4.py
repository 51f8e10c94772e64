#!/usr/bin/env python3
"""Per-XCD walker finishing times in a torch process (tools only; needs `make -C halo2_regex_amd/csrc stamps` and
HRX_LIB_PATH=.../libhrx_stamps.so): which XCDs lag in the fast and in the slow per-process state?"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("HRX_LIB_PATH", os.path.join(ROOT, "halo2_regex_amd", "csrc", "libhrx_stamps.so"))
sys.path.insert(0, ROOT)
import numpy as np, torch
import halo2_regex_amd as hra
from halo2_regex_amd import synth
B, N, M = int(os.environ.get("HRX_AB_BATCH", 65536)), int(os.environ.get("HRX_AB_LEN", 1023)), int(os.environ.get("HRX_AB_ROWS", 1024))
D_DIR = os.path.join(ROOT, "tests", "golden", "dfa")
rd = lambda f: open(os.path.join(D_DIR, f), "rb").read()
which = [int(c) for c in os.environ.get("HRX_AB_DEFS", "1")]          # e.g. HRX_AB_DEFS=23: regex2 + regex3
defs = [hra.RegexDefs(hra.AllstrRegexDef(rd("regex%d_test_lookup.txt" % k)), [hra.SubstrRegexDef(rd("substr%d_test_lookup.txt" % k))]) for k in which]
cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
dev = torch.device("cuda", 0)
stride = (N + 15) // 16 * 16
chars, lens = (synth.regex1_planted if which == [1] else synth.regex23_planted)(B, N, seed=0, stride=stride)
d_lens = torch.from_numpy(lens.astype(np.int32)).to(dev)
pm = hra.chars_to_position_major(torch.from_numpy(chars).to(dev))
out = cfg.alloc_outputs_position_major(B, dev)
step = lambda: cfg.witness_batch_position_major(pm, d_lens, out=out, chars_pm_stride=stride)
for _ in range(20): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
IT = max(5, int(100 * 65536 * 1024 / (B * M)))
for _ in range(IT): step()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / IT
n = 256 * 4 * 8
buf = (C.c_uint64 * n)()
hra.lib.hrx_debug_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t]
rc = hra.lib.hrx_debug_read_stamps(cfg._ctx, buf, n)
st = np.frombuffer(buf, dtype=np.uint64).reshape(256, 4, 8).astype(np.float64)
e = st[:, :, 4].min()
done = (st[:, :, 6] - e) / 100.0            # us after the first workgroup's entry, per (workgroup, pair)
start = (st[:, :, 5] - e) / 100.0
byx = [done[x::8].mean() for x in range(8)]
mhz = st[:, :, 3] / ((st[:, :, 6] - st[:, :, 5]) / 100.0)     # s_memtime ticks per microsecond of wall clock, per walker
print("s_memtime ticks per us by workgroup %% 8: %s" % " ".join("%.0f" % mhz[x::8].mean() for x in range(8)))
ntl = (M + 63) // 64 * max(1, B // 65536)
walk = st[:, :, 1] / ntl
print("tile-end ticks per tile: %.0f; groups total ticks per walker %.0f" % (st[:, :, 2].mean() / ntl, st[:, :, 3].mean()))
print("walk ticks per tile by workgroup %% 8: %s; input wait: %s" % (" ".join("%.0f" % walk[x::8].mean() for x in range(8)), " ".join("%.0f" % (st[x::8, :, 0].mean() / ntl) for x in range(8))))
print("%.1f us/launch (stamps build; memset between launches); walker-done us by workgroup %% 8: %s; first %.1f last %.1f; start max %.1f"
      % (us, " ".join("%.1f" % v for v in byx), done.min(), done.max(), start.max()))
