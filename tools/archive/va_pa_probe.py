#!/usr/bin/env python3
"""cfg 3 shape: the fast / slow state of a launch belongs to its RECORDS buffer (set_probe4.py).  To the virtual range or to the
physical memory?  K reserved ranges x K sets of physical chunks (HIP virtual memory management through ctypes), every
combination mapped and timed.  (tools only)"""
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
prop = Prop(); prop.type = 1; prop.loc.type = 1; prop.loc.id = 0       # hipMemAllocationTypePinned, hipMemLocationTypeDevice
acc = Acc(); acc.loc.type = 1; acc.loc.id = 0; acc.flags = 3           # hipMemAccessFlagsProtReadWrite
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
CH = int(os.environ.get("CHUNK_MIB", "16")) << 20
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = RB // CH
vas, pas = [], []
for s in range(K):
    p = C.c_void_p(); ck(hip.hipMemAddressReserve(C.byref(p), C.c_size_t(RB), C.c_size_t(CH), None, C.c_ulonglong(0)), "reserve"); vas.append(p.value)
    hs = []
    for c in range(n):
        h = C.c_void_p(); ck(hip.hipMemCreate(C.byref(h), C.c_size_t(CH), C.byref(prop), C.c_ulonglong(0)), "create"); hs.append(h)
    pas.append(hs)
    junk = torch.empty((100 + 300 * s) << 20, dtype=torch.uint8, device=dev)
class Ext:
    def __init__(self, ptr, nbytes): self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
def timeit(rec, k=12):
    for _ in range(2): cfg.witness_batch_position_major(pm0, d_lens, out=(rec, m0, st), chars_pm_stride=2048)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): cfg.witness_batch_position_major(pm0, d_lens, out=(rec, m0, st), chars_pm_stride=2048)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
for rnd in range(2):
    print("round %d — rows: virtual range, columns: physical set; us per launch" % rnd)
    for v in range(K):
        row = []
        for p in range(K):
            for c in range(n): ck(hip.hipMemMap(C.c_void_p(vas[v] + c * CH), C.c_size_t(CH), C.c_size_t(0), pas[p][c], C.c_ulonglong(0)), "map")
            ck(hip.hipMemSetAccess(C.c_void_p(vas[v]), C.c_size_t(RB), C.byref(acc), C.c_size_t(1)), "access")
            rec = torch.as_tensor(Ext(vas[v], RB), device=dev).view(torch.int32)
            row.append(timeit(rec))
            del rec
            torch.cuda.synchronize()
            ck(hip.hipMemUnmap(C.c_void_p(vas[v]), C.c_size_t(RB)), "unmap")
        print("  va %#x: " % vas[v] + " ".join("%7.1f" % x for x in row), flush=True)
