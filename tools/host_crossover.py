#!/usr/bin/env python3
"""Where the native host walk (csrc/hrx_host_walk.cpp, one host thread) and the device path (H2D + batch kernel + D2H +
synchronise) of hrx_witness_batch_host cross over: microseconds per call against the batch size, regex1 + substr1, M = 1024
(and D = 2, regex1 + regex2).  The library's default threshold (HRX_DEFAULT_HOST_THRESHOLD rows) is read off this table
(NOTES_MEASUREMENTS.md §7c).  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import halo2_regex_amd as hra
from halo2_regex_amd import synth
DD = os.path.join(ROOT, "tests", "golden", "dfa")
mk = lambda k: hra.RegexDefs(hra.AllstrRegexDef.read_from_text(os.path.join(DD, "regex%d_test_lookup.txt" % k)),
                             [hra.SubstrRegexDef.read_from_text(os.path.join(DD, "substr%d_test_lookup.txt" % k))])
M = 1024
for label, defs in (("D = 1 (regex1)", [mk(1)]), ("D = 2 (regex1 + regex2)", [mk(1), mk(2)])):
    cfgs = {}
    for name, flag in (("host", 0x10000000), ("device", 0x20000000)):   # kDbgForceHost / kDbgNoHost: read once per context
        os.environ["HRX_DEBUG_FLAGS"] = str(flag)
        cfgs[name] = hra.RegexVerifyConfig.configure(M, defs, device=0)
    os.environ.pop("HRX_DEBUG_FLAGS")
    print("%s, M = %d rows per string; us per hrx_witness_batch_host call (median of 15), output arrays reused" % (label, M))
    print("%8s %10s %12s %12s %8s" % ("strings", "rows", "host walk", "device path", "ratio"))
    chars_all, lens_all = synth.regex1_planted(4096, 1023, seed=0, stride=1024)
    for B in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096):
        chars, lens = chars_all[:B], lens_all[:B]
        res = {}
        for name, cfg in cfgs.items():
            out = cfg.witness_batch_host(chars, lens)
            ts = []
            for _ in range(15):
                t0 = time.perf_counter()
                cfg.witness_batch_host(chars, lens, out=out)
                ts.append(time.perf_counter() - t0)
            res[name] = sorted(ts)[7] * 1e6
        print("%8d %10d %12.1f %12.1f %8.2f" % (B, B * M, res["host"], res["device"], res["host"] / res["device"]))
