#!/usr/bin/env python3
"""Debug aid for the chunked launch: forced chunks of 4 tiles on a small batch, first differing row per string against the oracle."""
import os, sys
os.environ["HRX_DEBUG_FLAGS"] = str(int(os.environ.get("SPEC_FLAGS", "0x80"), 0) | 0x20000000)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import halo2_regex_amd as hra
from halo2_regex_amd import synth
from oracle_lib import OracleDefs, load_oracle, DFA_DIR
M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
which = sys.argv[3] if len(sys.argv) > 3 else "r1"
CFG = {"r1": [["regex1_test_lookup.txt", ["substr1_test_lookup.txt"]]],
       "r23": [["regex2_test_lookup.txt", ["substr2_test_lookup.txt"]], ["regex3_test_lookup.txt", ["substr3_test_lookup.txt"]]]}[which]
defs = [hra.RegexDefs(hra.AllstrRegexDef.read_from_text(os.path.join(DFA_DIR, a)), [hra.SubstrRegexDef.read_from_text(os.path.join(DFA_DIR, s)) for s in subs]) for a, subs in CFG]
cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
print(cfg.describe_launch(B, layout=3))
chars, lens = synth.reveal_stress(B, M, seed=40 + M)
lens[::3] = M
o = OracleDefs.from_files(load_oracle(), CFG)
orec, omsk, ost = o.witness_batch(chars, lens, M)
dev = torch.device("cuda", 0)
d_c = hra.chars_to_position_major(torch.from_numpy(chars).to(dev)); d_l = torch.from_numpy(lens.astype(np.int32)).to(dev)
rec, msk, st = cfg.witness_batch_position_major(d_c, d_l, chars_pm_stride=chars.shape[1])
torch.cuda.synchronize()
D = len(CFG)
r, m = hra.position_major_to_string_major(rec, msk, B, M, D)
r = r.cpu().numpy().view(np.uint32); m = m.cpu().numpy().view(np.uint16); s = st.cpu().numpy().view(np.uint64)
nbad = 0
for b in range(B):
    if s[b] != ost[b]:
        print("string %d len %d: status %x vs oracle %x" % (b, lens[b], s[b], ost[b])); nbad += 1; continue
    if ost[b] & 0xff: continue
    dr = np.nonzero((r[b] != orec[b]).any(axis=1))[0]
    dm = np.nonzero(m[b] != omsk[b])[0]
    if len(dr) or len(dm):
        nbad += 1
        if nbad <= 12:
            msg = "string %d len %d:" % (b, lens[b])
            if len(dr): msg += " records differ at rows %s.. (%d rows) got %s want %s" % (dr[:4], len(dr), [hex(x) for x in r[b, dr[0]]], [hex(x) for x in orec[b, dr[0]]])
            if len(dm): msg += " masked differ at rows %s.. (%d rows) got %x want %x" % (dm[:4], len(dm), m[b, dm[0]], omsk[b, dm[0]])
            print(msg)
print("bad strings: %d of %d" % (nbad, B))
