#!/usr/bin/env python3
"""cfg 5 launch, stamps build: when do the walkers of each XCD finish?  (workgroups are dealt to the 8 XCDs round-robin: XCD = blockIdx % 8)
  HRX_LIB_PATH=halo2_regex_amd/csrc/libhrx_stamps.so python3 tools/stamps_cfg5_xcd.py [B] [n]"""
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
outs = [cfg.alloc_outputs_position_major(B, dev) for _ in range(4)]
hra.lib.hrx_debug_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_size_t]
npairs = 1024
for rep in range(4):
    out = outs[rep]
    for i in range(2):
        cfg.witness_batch_position_major(d_c, d_l, out=out, chars_pm_stride=n)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (npairs * 16))()
    assert hra.lib.hrx_debug_read_stamps(cfg._ctx, buf, npairs * 16) == 0
    s = np.frombuffer(buf, dtype=np.uint64).reshape(npairs, 16).astype(np.int64)
    w_entry, w_start, w_end = s[:, 4], s[:, 5], s[:, 6]
    t0 = w_entry.min()
    end = (w_end - t0) * 0.01                       # us, 100 MHz wall clock
    wg = np.arange(npairs) // 4
    xcd = wg % 8
    print("output set %d: walker done, us after the launch's first wave: all %.1f .. %.1f (median %.1f)" % (rep, end.min(), end.max(), np.median(end)))
    print("   per XCD (median / max): " + "  ".join("%d: %.0f / %.0f" % (x, np.median(end[xcd == x]), end[xcd == x].max()) for x in range(8)))
    cu = wg // 8                                    # position of the workgroup within its XCD (32 CUs each)
    print("   per CU slot within the XCD, median over XCDs: " + " ".join("%.0f" % np.median(end[cu == c]) for c in range(0, 32, 4)) + " ...")
    print("   walk cycles per tile, per XCD: " + "  ".join("%d: %.0f" % (x, s[xcd == x, 1].mean() / ((n + 63) // 64)) for x in range(8)))
