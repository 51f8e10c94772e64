#!/usr/bin/env python3
"""cfg 3 shape: ONE physical 4-GiB allocation (hipMemCreate) mapped at virtual addresses of different alignment.  If the launch
time follows the alignment, the fast / slow state of a records buffer is the page-table fragment size the driver could use
for it (the largest 2^k that divides both the virtual and the physical address).  (tools only)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import halo2_regex_amd as hra
from halo2_regex_amd import synth
hip = C.CDLL("libamdhip64.so")
class Loc(C.Structure): _fields_ = [("type", C.c_int), ("id", C.c_int)]
class Prop(C.Structure): _fields_ = [("type", C.c_int), ("handle", C.c_int), ("loc", Loc), ("win32", C.c_void_p), ("comp", C.c_ubyte), ("rdma", C.c_ubyte), ("usage", C.c_ushort)]
class Acc(C.Structure): _fields_ = [("loc", Loc), ("flags", C.c_int)]
def ck(r, what):
    if r != 0: raise RuntimeError("%s -> %d" % (what, r))
prop = Prop(); prop.type = 1; prop.loc.type = 1; prop.loc.id = 0
acc = Acc(); acc.loc.type = 1; acc.loc.id = 0; acc.flags = 3
B, N, M = 262144, 2047, 2048
D_DIR = os.path.join(ROOT, "tests", "golden", "dfa")
rd = lambda f: open(os.path.join(D_DIR, f), "rb").read()
defs = [hra.RegexDefs(hra.AllstrRegexDef(rd("regex%d_test_lookup.txt" % k)), [hra.SubstrRegexDef(rd("substr%d_test_lookup.txt" % k))]) for k in (2, 3)]
cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
dev = torch.device("cuda", 0)
chars, lens = synth.regex23_planted(B, N, seed=0, stride=2048)
d_lens = torch.from_numpy(lens.astype(np.int32)).to(dev)
pm0 = hra.chars_to_position_major(torch.from_numpy(chars).to(dev))
RB = B * M * 2 * 4
m0 = torch.empty(B * M, dtype=torch.int16, device=dev); st = torch.empty(B, dtype=torch.int64, device=dev)
GIB = 1 << 30
CH = int(os.environ.get("CHUNK_MIB", "4096")) << 20
base = C.c_void_p(); ck(hip.hipMemAddressReserve(C.byref(base), C.c_size_t(16 * GIB), C.c_size_t(0), None, C.c_ulonglong(0)), "reserve")
hs = []
for c in range(RB // CH):
    h = C.c_void_p(); ck(hip.hipMemCreate(C.byref(h), C.c_size_t(CH), C.byref(prop), C.c_ulonglong(0)), "create"); hs.append(h)
class Ext:
    def __init__(self, ptr, nbytes): self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
def timeit(rec, k=12):
    for _ in range(2): cfg.witness_batch_position_major(pm0, d_lens, out=(rec, m0, st), chars_pm_stride=2048)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): cfg.witness_batch_position_major(pm0, d_lens, out=(rec, m0, st), chars_pm_stride=2048)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
top = (base.value + 8 * GIB) // (8 * GIB) * (8 * GIB)      # an 8-GiB aligned address inside the reservation
print("reservation %#x, chunk %d MiB" % (base.value, CH >> 20))
for rnd in range(2):
    for lg in (21, 22, 23, 24, 25, 26, 28, 30, 32, 33):
        va = top if lg == 33 else top + (1 << lg)               # exactly 2^lg aligned
        if va + RB > base.value + 16 * GIB: va = top - (1 << lg) if lg < 33 else top
        for c, h in enumerate(hs): ck(hip.hipMemMap(C.c_void_p(va + c * CH), C.c_size_t(CH), C.c_size_t(0), h, C.c_ulonglong(0)), "map")
        ck(hip.hipMemSetAccess(C.c_void_p(va), C.c_size_t(RB), C.byref(acc), C.c_size_t(1)), "access")
        rec = torch.as_tensor(Ext(va, RB), device=dev).view(torch.int32)
        t = timeit(rec)
        del rec; torch.cuda.synchronize()
        ck(hip.hipMemUnmap(C.c_void_p(va), C.c_size_t(RB)), "unmap")
        print("  va %#x (2^%d aligned): %7.1f us" % (va, lg, t), flush=True)
