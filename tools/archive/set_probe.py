#!/usr/bin/env python3
"""cfg 3 shape (regex2+regex3, 262144 x 2048 B, 5.9 GB per buffer set): launch time per BUFFER SET inside one process (tools only).
Fresh processes are bimodal (0.97 vs 1.14 ms); is the state the process's or the buffers' (their physical placement)?
usage: set_probe.py [nsets] [shuffle MiB ...]   — shuffle: junk allocations made (and kept) between the sets"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import halo2_regex_amd as hra
from halo2_regex_amd import synth
B, N, M = 262144, 2047, 2048
D_DIR = os.path.join(ROOT, "tests", "golden", "dfa")
rd = lambda f: open(os.path.join(D_DIR, f), "rb").read()
defs = [hra.RegexDefs(hra.AllstrRegexDef(rd("regex%d_test_lookup.txt" % k)), [hra.SubstrRegexDef(rd("substr%d_test_lookup.txt" % k))]) for k in (2, 3)]
cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
dev = torch.device("cuda", 0)
nsets = int(sys.argv[1]) if len(sys.argv) > 1 else 6
chars, lens = synth.regex23_planted(B, N, seed=0, stride=2048)
d_lens = torch.from_numpy(lens.astype(np.int32)).to(dev)
h = torch.from_numpy(chars)
sets, junk = [], []
for s in range(nsets):
    pm = hra.chars_to_position_major(h.to(dev))
    out = cfg.alloc_outputs_position_major(B, dev)
    sets.append((pm, out))
    junk.append(torch.empty((37 + 64 * s) << 20, dtype=torch.uint8, device=dev))
K = int(os.environ.get('PROBE_K', '20'))
def timeit(pm, out, k=K):
    for _ in range(3): cfg.witness_batch_position_major(pm, d_lens, out=out, chars_pm_stride=2048)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): cfg.witness_batch_position_major(pm, d_lens, out=out, chars_pm_stride=2048)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
for rnd in range(int(os.environ.get('PROBE_ROUNDS', '3'))):
    print("round %d: " % rnd + "  ".join("%.3f" % timeit(pm, out) for pm, out in sets))
if os.environ.get('PROBE_ROUNDS'): sys.exit(0)
# mixed: input of set i with outputs of set j
print("in0->out*: " + "  ".join("%.3f" % timeit(sets[0][0], out) for _, out in sets))
print("in*->out0: " + "  ".join("%.3f" % timeit(pm, sets[0][1]) for pm, _ in sets))
print("ptrs: " + "  ".join("%#x/%#x/%#x" % (pm.data_ptr(), out[0].data_ptr(), out[1].data_ptr()) for pm, out in sets))
