"""PCIe-inclusive rate of the host-buffer entry point (hrx_witness_batch_host) on the bench workload: NOTES_MEASUREMENTS.md §4.4.
Never the bench's `value`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import halo2_regex_amd as hra
from halo2_regex_amd import synth
D = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dfa")
defs = [hra.RegexDefs(hra.AllstrRegexDef.read_from_text(os.path.join(D, "regex1_test_lookup.txt")),
                      [hra.SubstrRegexDef.read_from_text(os.path.join(D, "substr1_test_lookup.txt"))])]
cfg = hra.RegexVerifyConfig.configure(1024, defs, device=0)
chars, lens = synth.regex1_planted(65536, 1023, seed=0, stride=1024)
cfg.witness_batch_host(chars, lens)
def rate(label, **kw):
    t0 = time.perf_counter()
    for _ in range(3):
        cfg.witness_batch_host(chars, lens, **kw)
    dt = (time.perf_counter() - t0) / 3
    print("hrx_witness_batch_host, 65536 x 1024-byte strings, %s: %.1f ms per call, %.3e rows/s, %.1f GB/s over the 7 B/row"
          % (label, dt * 1e3, lens.sum() / dt, 7 * lens.sum() / dt / 1e9))
rate("pageable host buffers, fresh output arrays every call")
out = cfg.witness_batch_host(chars, lens)
rate("pageable host buffers, output arrays reused", out=out)
