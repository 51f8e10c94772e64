#!/usr/bin/env python3
"""The bench line's launches timed PER BUFFER SET (events around every eager launch, the sets in rotation so that every byte comes from / goes to HBM): is a slow process slow on
some sets only (sub-buffer pairings inside a good arena pair) or on all of them?  python3 tools/per_set_times.py [--config ... like bench.py]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

def main():
    args = bench.parse_args(sys.argv[1:])
    import numpy as np, torch
    import halo2_regex_amd as hra
    dev = torch.device("cuda", 0)
    names, label, alphabet, gen, planted = bench.workload(args)
    M, n, B = args.rows, args.n, args.batch
    stride = (max(n, 1) + 15) // 16 * 16
    defs = [hra.RegexDefs(hra.AllstrRegexDef(a), [hra.SubstrRegexDef(t) for t in subs]) for a, subs in names]
    cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
    chars, lens = gen(B, n, seed=0, stride=stride)
    d_lens = torch.from_numpy(lens.astype(np.int32)).to(dev)
    d_chars = torch.from_numpy(chars).to(dev)
    nsets = 8
    shift = (B // nsets + 37) % B
    sets, reps = [], []
    for k in range(nsets):
        c = d_chars if k == 0 else torch.roll(d_chars, shifts=k * shift, dims=0)
        l = d_lens if k == 0 else torch.roll(d_lens, shifts=k * shift, dims=0)
        sets.append((hra.chars_to_position_major(c), l, cfg.alloc_outputs_position_major(B, dev)))
        reps.append(cfg.last_placement_report())
    torch.cuda.synchronize()
    launch = lambda i: cfg.witness_batch_position_major(sets[i][0], sets[i][1], out=sets[i][2], chars_pm_stride=stride)
    for r in range(30):
        for i in range(nsets):
            launch(i)
    torch.cuda.synchronize()
    R = 60
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nsets)] for _ in range(R)]
    for r in range(R):
        for i in range(nsets):
            ev[r][i][0].record(); launch(i); ev[r][i][1].record()
    torch.cuda.synchronize()
    t = np.array([[ev[r][i][0].elapsed_time(ev[r][i][1]) * 1e3 for i in range(nsets)] for r in range(R)])
    rep = reps[0]
    print("arena pair: steps %s best %.0f GB/s ref %.0f" % (rep.get("steps"), rep.get("best_gbs", 0), rep.get("ref_gbs", 0)))
    print("records at +MiB:", " ".join("%5d" % ((s[2][0].data_ptr() - sets[0][2][0].data_ptr()) >> 20) for s in sets))
    print("masked  at +MiB:", " ".join("%5d" % ((s[2][1].data_ptr() - sets[0][2][1].data_ptr()) >> 20) for s in sets))
    print("us per launch, median over %d rotations, per set:" % R, " ".join("%.1f" % x for x in np.median(t, axis=0)), " all: %.2f" % np.median(t))

if __name__ == "__main__":
    main()
