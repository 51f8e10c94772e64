#!/usr/bin/env python3
"""cfg 4 (def-parallel kernel), stamps build: how far apart do the groups' write positions drift over a launch?  Every group's combiner stamps the wall
clock when it reaches each eighth of its rows; the spread of those times over the groups, divided by the time a tile takes, is the width of the write
front in tiles.
  HRX_LIB_PATH=halo2_regex_amd/csrc/libhrx_stamps.so python3 tools/front_width.py [B] [M] ..."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import halo2_regex_amd as hra
from halo2_regex_amd import synth
import bench
assert hasattr(hra.lib, "hrx_debug_read_stamps"), "load libhrx_stamps.so through HRX_LIB_PATH"
hra.lib.hrx_debug_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_size_t]
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
names = bench.workload(bench.parse_args(["--config", "headers3"]))[0]
defs = [hra.RegexDefs(hra.AllstrRegexDef(a), [hra.SubstrRegexDef(t) for t in subs]) for a, subs in names]
for M in [int(x) for x in sys.argv[2:]] or [8192, 32768]:
    cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
    desc = cfg.describe_launch(B, layout=3)
    print("== %d x %d: %s" % (B, M, desc))
    base_c, base_l = synth.headers_planted(2048, M - 1, seed=3, stride=M)
    d_c = torch.from_numpy(base_c).to(dev)
    d_c = torch.cat([torch.roll(d_c, shifts=131 * j, dims=0) for j in range(B // 2048)])
    d_l = torch.from_numpy(np.tile(base_l, B // 2048).astype(np.int32)).to(dev)
    for j in range(B // 2048):
        d_l[j * 2048:(j + 1) * 2048] = torch.roll(torch.from_numpy(base_l.astype(np.int32)).to(dev), shifts=131 * j)
    d_c = hra.chars_to_position_major(d_c)
    outs = [cfg.alloc_outputs_position_major(B, dev) for _ in range(3)]
    ngroups = B // 64
    ntiles = M // 64
    for rep in range(3):
        for i in range(2):
            cfg.witness_batch_position_major(d_c, d_l, out=outs[rep], chars_pm_stride=M)
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * (ngroups * 16))()
        assert hra.lib.hrx_debug_read_stamps(cfg._ctx, buf, ngroups * 16) == 0
        s = np.frombuffer(buf, dtype=np.uint64).reshape(ngroups, 16).astype(np.int64)[:, :9]
        t0 = s[:, 0].min()
        us = (s - t0) * 0.01
        per_tile = (us[:, 8] - us[:, 0]).mean() / ntiles
        wg = np.arange(ngroups) // 2
        xcd = wg % 8
        print(" output set %d: launch %.0f us, %.2f us per tile; front width (tiles) at each eighth of the rows, max-min | p95-p5:" % (rep, us[:, 8].max(), per_tile))
        print("   " + "  ".join("%d/8: %.0f | %.0f" % (k, (us[:, k].max() - us[:, k].min()) / per_tile, (np.percentile(us[:, k], 95) - np.percentile(us[:, k], 5)) / per_tile) for k in range(1, 9)))
        print("   finish per XCD, us (median): " + " ".join("%d: %.0f" % (x, np.median(us[xcd == x, 8])) for x in range(8)))
        print("   start spread %.1f us" % (us[:, 0].max() - us[:, 0].min()))
        if os.environ.get("FRONT_DUMP"):
            np.save(os.path.join(os.environ["FRONT_DUMP"], "front_%d_%d_set%d.npy" % (B, M, rep)), us)
    del outs, cfg
