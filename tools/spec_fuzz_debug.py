#!/usr/bin/env python3
"""one seed of test_fuzz_chunked_launch_on_random_definitions, with the differences decoded"""
import os, sys
os.environ["HRX_DEBUG_FLAGS"] = str((0 if int(sys.argv[1]) >= 100000 else 0x80) | 0x20000000)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import halo2_regex_amd as hra
from oracle_lib import OracleDefs, load_oracle, decode_status
import test_parity_gpu as T
seed = int(sys.argv[1])
rng = np.random.default_rng(5000 + seed)
D = int(rng.integers(1, 4))
defs_t = T._random_defs(rng, D, False)
natural = seed >= 100000
M = int(rng.choice([4096, 5120, 8192])) if natural else int(rng.choice([512, 768, 1024, 1280, 2048]))
B = int(rng.choice([1, 64, 65, 130])) if natural else int(rng.choice([1, 64, 65, 200, 333]))
stride = M
common = defs_t[0][2]
for _, _, a in defs_t[1:]:
    common = np.intersect1d(common, a)
alpha = np.unique(np.concatenate([a for _, _, a in defs_t]))
pool = common if len(common) >= 2 else alpha
chars = pool[rng.integers(0, len(pool), size=(B, stride))].astype(np.uint8)
lens = rng.integers(0, M + 1, size=B).astype(np.uint32)
lens[rng.random(B) < 0.3] = M
lens[rng.random(B) < 0.1] = 256 * int(rng.integers(1, M // 256 + 1))
lens[rng.random(B) < 0.03] = M + 3
for b in np.nonzero(rng.random(B) < 0.1)[0]:
    chars[b, int(rng.integers(0, stride))] = 0
defs = [hra.RegexDefs(hra.AllstrRegexDef(a), [hra.SubstrRegexDef(t) for t in subs]) for a, subs, _ in defs_t]
cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
print("D", D, "M", M, "B", B, "states", [a.splitlines()[2] for a, _, _ in defs_t], cfg.describe_launch(B, layout=3)[:120])
orec, omsk, ost = OracleDefs(load_oracle(), [(a, subs) for a, subs, _ in defs_t]).witness_batch(chars, lens, M)
dev = torch.device("cuda", 0)
rec, msk, st = cfg.witness_batch_position_major(hra.chars_to_position_major(torch.from_numpy(chars).to(dev)), torch.from_numpy(lens.astype(np.int32)).to(dev), chars_pm_stride=stride)
torch.cuda.synchronize()
r, m = hra.position_major_to_string_major(rec, msk, B, M, D)
r = r.cpu().numpy().view(np.uint32); m = m.cpu().numpy().view(np.uint16); s = st.cpu().numpy().view(np.uint64)
nb = 0
for b in range(B):
    if s[b] != ost[b]:
        nb += 1
        if nb <= 8: print("string %d len %d: status got %s want %s" % (b, lens[b], decode_status(s[b]), decode_status(ost[b])))
        continue
    if ost[b] & 0xff: continue
    dr = np.nonzero((r[b] != orec[b]).any(axis=1))[0]; dm = np.nonzero(m[b] != omsk[b])[0]
    if len(dr) or len(dm):
        nb += 1
        if nb <= 8: print("string %d len %d: records differ at %s (%d), masked at %s (%d)" % (b, lens[b], dr[:4], len(dr), dm[:4], len(dm)), [hex(x) for x in r[b, dr[0]]] if len(dr) else "", [hex(x) for x in orec[b, dr[0]]] if len(dr) else "")
print("bad", nb, "of", B)
