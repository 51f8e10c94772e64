#!/usr/bin/env python3
"""hrx_witness_batch_host call by call (HRX_HOST_TRACE=1 prints what each call did): the default route's measuring calls, then steady state; then the forced routes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HRX_HOST_TRACE", "1")
import numpy as np
import halo2_regex_amd as hra
from halo2_regex_amd import synth
D_DIR = os.path.join(ROOT, "tests", "golden", "dfa")
rd = lambda f: open(os.path.join(D_DIR, f), "rb").read()
B, n, M = 65536, 1023, 1024
defs = [hra.RegexDefs(hra.AllstrRegexDef(rd("regex1_test_lookup.txt")), [hra.SubstrRegexDef(rd("substr1_test_lookup.txt"))])]
cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
chars, lens = synth.regex1_planted(B, n, seed=0, stride=1024)
out = (np.empty((B, M, 1), np.uint32), np.empty((B, M), np.uint16), np.empty(B, np.uint64))
for name, route, calls in (("auto", hra.HOST_ROUTE_AUTO, int(sys.argv[1]) if len(sys.argv) > 1 else 16), ("device", hra.HOST_ROUTE_DEVICE, 6), ("host", hra.HOST_ROUTE_HOST, 6)):
    cfg.set_option(hra.OPT_HOST_ROUTE, route)
    for i in range(calls):
        t0 = time.perf_counter()
        cfg.witness_batch_host(chars, lens, out=out)
        ms = (time.perf_counter() - t0) * 1e3
        r = cfg.host_route_report()
        print("%s call %2d: %7.2f ms  did %s  dev %5d strings %.2f ms  host %5d strings %.2f ms on %d threads" % (name, i, ms, {0: "split", 1: "device", 2: "host"}[r["route"]], r["device_strings"], r["device_ms"],
                                                                                                                r["host_strings"], r["host_ms"], r["host_threads"]), flush=True)
