import os, sys, time
sys.path.insert(0, "/root/repo")
import bench, numpy as np, torch
import halo2_regex_amd as hra
args = bench.parse_args([])
dev = torch.device("cuda", 0)
names, label, alphabet, gen, planted = bench.workload(args)
M, n, B = args.rows, args.n, args.batch
stride = (n + 15) // 16 * 16
defs = [hra.RegexDefs(hra.AllstrRegexDef(a), [hra.SubstrRegexDef(t) for t in subs]) for a, subs in names]
cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
chars, lens = gen(B, n, seed=0, stride=stride)
d_l = torch.from_numpy(lens.astype(np.int32)).to(dev)
d_c = hra.chars_to_position_major(torch.from_numpy(chars).to(dev))
out = cfg.alloc_outputs_position_major(B, dev)
small_B = 64
d_l2 = d_l[:small_B].contiguous(); d_c2 = hra.chars_to_position_major(torch.from_numpy(chars[:small_B]).to(dev)); out2 = cfg.alloc_outputs_position_major(small_B, dev)
for _ in range(20): cfg.witness_batch_position_major(d_c2, d_l2, out=out2, chars_pm_stride=stride)
torch.cuda.synchronize()
N = 2000
t = time.perf_counter()
for _ in range(N): cfg.witness_batch_position_major(d_c2, d_l2, out=out2, chars_pm_stride=stride)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("tiny batch (64 strings): host %.1f us per call issued, %.1f us per call incl. drain" % ((t1 - t) / N * 1e6, (t2 - t) / N * 1e6))
for _ in range(20): cfg.witness_batch_position_major(d_c, d_l, out=out, chars_pm_stride=stride)
torch.cuda.synchronize()
N = 500
t = time.perf_counter()
for _ in range(N): cfg.witness_batch_position_major(d_c, d_l, out=out, chars_pm_stride=stride)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("bench batch: host %.1f us per call issued, %.1f us per call incl. drain (one buffer set)" % ((t1 - t) / N * 1e6, (t2 - t) / N * 1e6))
