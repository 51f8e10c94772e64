#!/usr/bin/env python3
"""cfg 3 shape: launch time by HOW the records buffer was allocated — hipMalloc, hipExtMallocWithFlags(contiguous), and hipMalloc
after the free memory has been churned.  (tools only)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import halo2_regex_amd as hra
from halo2_regex_amd import synth
hip = C.CDLL("libamdhip64.so")
def ck(r, what):
    if r != 0: raise RuntimeError("%s -> %d" % (what, r))
B, N, M = 262144, 2047, 2048
D_DIR = os.path.join(ROOT, "tests", "golden", "dfa")
rd = lambda f: open(os.path.join(D_DIR, f), "rb").read()
defs = [hra.RegexDefs(hra.AllstrRegexDef(rd("regex%d_test_lookup.txt" % k)), [hra.SubstrRegexDef(rd("substr%d_test_lookup.txt" % k))]) for k in (2, 3)]
cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
dev = torch.device("cuda", 0)
chars, lens = synth.regex23_planted(B, N, seed=0, stride=2048)
d_lens = torch.from_numpy(lens.astype(np.int32)).to(dev)
pm0 = hra.chars_to_position_major(torch.from_numpy(chars).to(dev))
r0, m0, st = cfg.alloc_outputs_position_major(B, dev)
RB = r0.numel() * 4
del r0
class Ext:
    def __init__(self, ptr, nbytes): self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
def timeit(ptr, k=12):
    rec = torch.as_tensor(Ext(ptr, RB), device=dev).view(torch.int32)
    for _ in range(2): cfg.witness_batch_position_major(pm0, d_lens, out=(rec, m0, st), chars_pm_stride=2048)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): cfg.witness_batch_position_major(pm0, d_lens, out=(rec, m0, st), chars_pm_stride=2048)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
def alloc(kind, nbytes=None):
    p = C.c_void_p()
    nbytes = RB if nbytes is None else nbytes
    if kind == "malloc": ck(hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)), "hipMalloc")
    else: ck(hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(nbytes), C.c_uint({"contig": 4, "uncached": 3, "fine": 1}[kind])), kind)
    return p.value
for kind in os.environ.get("KINDS", "malloc contig malloc contig uncached fine").split():
    ps = [alloc(kind) for _ in range(int(os.environ.get("NBUF", "6")))]
    print("%-8s " % kind + " ".join("%7.1f" % timeit(p) for p in ps) + "   " + " ".join("%#x" % p for p in ps), flush=True)
    for p in ps: ck(hip.hipFree(C.c_void_p(p)), "free")
# sizes just above 4 GiB: does the size class of the allocation matter?
for extra in [int(x) << 20 for x in os.environ.get("EXTRAS", "0 2 64 1024 4096").split()]:
    ps = [alloc("malloc", RB + extra) for _ in range(4)]
    print("malloc +%4d MiB " % (extra >> 20) + " ".join("%7.1f" % timeit(p) for p in ps), flush=True)
    for p in ps: ck(hip.hipFree(C.c_void_p(p)), "free")
