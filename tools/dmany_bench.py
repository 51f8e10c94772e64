import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import halo2_regex_amd as hra
from halo2_regex_amd import synth
from oracle_lib import DFA_DIR
import test_parity_gpu as t
dev = torch.device("cuda", 0)
B, M = 65536, 1024
base_c, base_l = synth.headers_planted(4096, M - 1, seed=3, stride=M)
d_c = torch.from_numpy(base_c).to(dev); d_c = torch.cat([torch.roll(d_c, shifts=131 * j, dims=0) for j in range(B // 4096)])
d_l0 = torch.from_numpy(base_l.astype(np.int32)).to(dev); d_l = torch.cat([torch.roll(d_l0, shifts=131 * j, dims=0) for j in range(B // 4096)])
d_c = hra.chars_to_position_major(d_c); rows = int(d_l.sum())
for nm, names in (("D8", t.CFG_D8), ("D13", t.CFG_D13), ("D16", t.CFG_D16), ("D32", t.CFG_D32)):
    defs = [hra.RegexDefs(hra.AllstrRegexDef.read_from_text(os.path.join(DFA_DIR, a)), [hra.SubstrRegexDef.read_from_text(os.path.join(DFA_DIR, s)) for s in subs]) for a, subs in names]
    cfg = hra.RegexVerifyConfig.configure(M, defs, device=0); D = len(names)
    outs = [cfg.alloc_outputs_position_major(B, dev) for _ in range(3)]
    for i in range(3): cfg.witness_batch_position_major(d_c, d_l, out=outs[i], chars_pm_stride=M)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(9): cfg.witness_batch_position_major(d_c, d_l, out=outs[i % 3], chars_pm_stride=M)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 9
    print("%s: %.3f ms  frac %.3f   %s" % (nm, ms, rows * (4 * D + 3) / (ms * 1e-3) / 8e12, cfg.describe_launch(B, layout=3)[:90]))
    del outs, cfg
