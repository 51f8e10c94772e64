"""hrx_witness_batch_host call by call (output arrays reused): is the call's duration the same in every process and every call?  python3 tools/host_path_modes.py [calls]
Prints per-call ms, a plain copy-out of as many bytes in between, and which CPUs the process may run on."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import halo2_regex_amd as hra
from halo2_regex_amd import synth
D = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dfa")
defs = [hra.RegexDefs(hra.AllstrRegexDef.read_from_text(os.path.join(D, "regex1_test_lookup.txt")),
                      [hra.SubstrRegexDef.read_from_text(os.path.join(D, "substr1_test_lookup.txt"))])]
cfg = hra.RegexVerifyConfig.configure(1024, defs, device=0)
chars, lens = synth.regex1_planted(65536, 1023, seed=0, stride=1024)
out = cfg.witness_batch_host(chars, lens)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ts = []
for i in range(n):
    t0 = time.perf_counter(); cfg.witness_batch_host(chars, lens, out=out); ts.append((time.perf_counter() - t0) * 1e3)
dev = torch.device("cuda", 0)
d_rec = torch.empty(out[0].nbytes, dtype=torch.uint8, device=dev); d_msk = torch.empty(out[1].nbytes, dtype=torch.uint8, device=dev)
h_rec = torch.from_numpy(out[0].view(np.uint8).reshape(-1)); h_msk = torch.from_numpy(out[1].view(np.uint8).reshape(-1))
cs = []
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter(); h_rec.copy_(d_rec); h_msk.copy_(d_msk); torch.cuda.synchronize(); cs.append((time.perf_counter() - t0) * 1e3)
print("pid %d cpus %s  host path ms: %s | plain copy-out ms: %s" % (os.getpid(), len(os.sched_getaffinity(0)), " ".join("%.1f" % t for t in ts), " ".join("%.1f" % t for t in cs)))
