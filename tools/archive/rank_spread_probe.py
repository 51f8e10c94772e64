#!/usr/bin/env python3
"""cfg 3 shape with buffers whose physical chunks come from different thirds of the device memory (tools only).
tools/halves_probe.cpp: two write streams run at 6.3 TB/s inside one region of the physical address space and at 7.5 TB/s
across two; the regions look like the three 96-GiB ranks of the 12-high HBM stacks, and a fresh process allocates top-down,
i.e. everything from ONE of them.  Here chunks are created (HIP virtual memory management) while spacer allocations of
96 GiB push the allocator into the next region, then mapped round-robin into one virtual range."""
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
st = torch.empty(B, dtype=torch.int64, device=dev)
GIB = 1 << 30
RB, MB, CB = B * M * 2 * 4, B * M * 2, B * M
CH = int(os.environ.get("CHUNK_MIB", "2")) << 20
class Ext:
    def __init__(self, ptr, nbytes): self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
def create(n):
    hs = []
    for _ in range(n):
        h = C.c_void_p(); ck(hip.hipMemCreate(C.byref(h), C.c_size_t(CH), C.byref(prop), C.c_ulonglong(0)), "create"); hs.append(h)
    return hs
def spacer(gib):
    p = C.c_void_p(); ck(hip.hipMalloc(C.byref(p), C.c_size_t(int(gib * GIB))), "spacer"); return p
def build(nbytes, pools, pattern):
    """a virtual range of nbytes whose chunk c comes from pools[pattern(c)]"""
    n = (nbytes + CH - 1) // CH
    va = C.c_void_p(); ck(hip.hipMemAddressReserve(C.byref(va), C.c_size_t(n * CH), C.c_size_t(0), None, C.c_ulonglong(0)), "reserve")
    for c in range(n):
        ck(hip.hipMemMap(C.c_void_p(va.value + c * CH), C.c_size_t(CH), C.c_size_t(0), pools[pattern(c)].pop(), C.c_ulonglong(0)), "map")
    ck(hip.hipMemSetAccess(va, C.c_size_t(n * CH), C.byref(acc), C.c_size_t(1)), "access")
    return torch.as_tensor(Ext(va.value, nbytes), device=dev)
def timeit(pm, rec, msk, k=12):
    out = (rec.view(torch.int32), msk.view(torch.int16), st)
    for _ in range(3): cfg.witness_batch_position_major(pm, d_lens, out=out, chars_pm_stride=2048)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): cfg.witness_batch_position_major(pm, d_lens, out=out, chars_pm_stride=2048)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
free, total = torch.cuda.mem_get_info()
print("free %.1f of %.1f GiB, chunk %d MiB" % (free / GIB, total / GIB, CH >> 20))
need = 3 * ((RB + MB + CB) // CH + 3)            # chunks per region: enough for three variants
pools = [create(need)]
s1 = spacer(96 - need * CH / GIB); pools.append(create(need))
s2 = spacer(96 - need * CH / GIB); pools.append(create(need))
hip.hipFree(s1); hip.hipFree(s2)
plain = (torch.empty(RB, dtype=torch.uint8, device=dev), torch.empty(MB, dtype=torch.uint8, device=dev))
print("plain torch buffers:                              %7.1f us" % timeit(pm0, *plain))
rec = build(RB, pools, lambda c: 0); msk = build(MB, pools, lambda c: 1)
print("records in region 0, masked rows in region 1:     %7.1f us" % timeit(pm0, rec, msk))
rec = build(RB, pools, lambda c: c % 3); msk = build(MB, pools, lambda c: c % 3)
print("records and masked rows round-robin over 3:       %7.1f us" % timeit(pm0, rec, msk))
pm = build(CB, pools, lambda c: c % 3); pm.copy_(pm0)
print("... and the input:                                %7.1f us" % timeit(pm, rec, msk))
rec2 = build(RB, pools, lambda c: c % 2); msk2 = build(MB, pools, lambda c: c % 2)
print("records and masked rows round-robin over 2:       %7.1f us" % timeit(pm0, rec2, msk2))
