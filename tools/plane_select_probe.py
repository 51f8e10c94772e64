#!/usr/bin/env python3
"""Which property of a set of record planes + masked rows predicts the launch time?  Allocates P plane candidates and Q masked-row candidates one after the other (what
hrx_alloc_output_planes does), measures every pairing (hrx_probe_write_pair), then times the REAL launch (and the no-compute pass) over EVERY combination of D planes x 1 masked
buffer and prints them sorted, with the pairings of each combination — the data the allocator's selection rule is read off.

  python tools/plane_select_probe.py --config headers3 --batch 32768 --len 32767 --rows 32768 --distinct 4096 --planes 8 --masked 4
"""
import argparse
import itertools
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="headers3")
    ap.add_argument("--batch", type=int, default=32768)
    ap.add_argument("--len", type=int, default=32767, dest="n")
    ap.add_argument("--rows", type=int, default=32768)
    ap.add_argument("--distinct", type=int, default=4096)
    ap.add_argument("--planes", type=int, default=8)
    ap.add_argument("--masked", type=int, default=4)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--max-combos", type=int, default=400)
    ap.add_argument("--spacer-gib", type=float, default=0.0, help="allocate (and keep) this much before the candidates: where in the device memory the pool lies")
    a = ap.parse_args()
    import numpy as np
    import torch
    import bench
    import halo2_regex_amd as hra
    wa = bench.parse_args(["--config", a.config, "--batch", str(a.batch), "--len", str(a.n), "--rows", str(a.rows)])
    names, label, alphabet, gen, planted = bench.workload(wa)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    D, M, n, B = len(names), a.rows, a.n, a.batch
    stride = (max(n, 1) + 15) // 16 * 16
    defs = [hra.RegexDefs(hra.AllstrRegexDef(x), [hra.SubstrRegexDef(t) for t in subs]) for x, subs in names]
    nd = min(B, a.distinct) if a.distinct > 0 else B
    sb = (nd // 7 + 11) % nd if nd < B else 0
    chars, lens = gen(nd, n, seed=0, stride=stride)
    d_lens = torch.from_numpy(lens.astype(np.int32)).to(dev)
    d_chars = torch.from_numpy(chars).to(dev)
    if nd < B:
        nblk = (B + nd - 1) // nd
        d_chars = torch.cat([torch.roll(d_chars, shifts=j * sb, dims=0) if j else d_chars for j in range(nblk)])[:B].contiguous()
        d_lens = torch.cat([torch.roll(d_lens, shifts=j * sb, dims=0) if j else d_lens for j in range(nblk)])[:B].contiguous()
    rows = int(d_lens.sum(dtype=torch.int64))
    alg = rows * (1 + 4 * D + 2)
    c_pm = hra.chars_to_position_major(d_chars)
    del d_chars
    torch.cuda.empty_cache()
    cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
    q4, q8 = (M + 3) // 4, (M + 7) // 8
    spacer = torch.empty(int(a.spacer_gib * (1 << 30)), dtype=torch.uint8, device=dev) if a.spacer_gib > 0 else None
    P = [torch.empty(q4 * B * 4, dtype=torch.int32, device=dev) for _ in range(a.planes)]
    Q = [torch.empty(q8 * B * 8, dtype=torch.int16, device=dev) for _ in range(a.masked)]
    st = torch.empty(B, dtype=torch.int64, device=dev)
    print("%s  D=%d  %d x %d B  algorithmic %.3f GB  | %s" % (label, D, B, stride, alg * 1e-9, cfg.describe_launch(B, layout=3 | hra.LAYOUT_RECORD_PLANES).split(" lds=")[0]))
    for i, p in enumerate(P):
        print("P[%d] %#x" % (i, p.data_ptr()))
    for i, q in enumerate(Q):
        print("Q[%d] %#x" % (i, q.data_ptr()))
    pp = np.zeros((a.planes, a.planes))
    pq = np.zeros((a.masked, a.planes))
    for i in range(a.planes):
        for j in range(i + 1, a.planes):
            pp[i, j] = pp[j, i] = cfg.probe_write_pair(P[i], P[j])
    for i in range(a.masked):
        for j in range(a.planes):
            pq[i, j] = cfg.probe_write_pair(Q[i], P[j])
    print("plane x plane pairings (GB/s / 10):")
    for i in range(a.planes):
        print("  %2d: %s" % (i, " ".join("%4.0f" % (pp[i, j] / 10) for j in range(a.planes))))
    print("masked x plane pairings:")
    for i in range(a.masked):
        print("  m%d: %s" % (i, " ".join("%4.0f" % (pq[i, j] / 10) for j in range(a.planes))))

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.steps

    combos = [(c, m) for c in itertools.combinations(range(a.planes), D) for m in range(a.masked)]
    if len(combos) > a.max_combos:
        rng = np.random.default_rng(0)
        combos = [combos[i] for i in sorted(rng.choice(len(combos), a.max_combos, replace=False))]
    res = []
    for c, m in combos:
        out = ([P[i] for i in c], Q[m], st)
        ms = timed(lambda: cfg.witness_batch_planes(c_pm, d_lens, out=out, chars_pm_stride=stride))
        ps = timed(lambda: cfg.traffic_pass_planes(c_pm, B, out, stride))
        pairs = [pp[x, y] for x, y in itertools.combinations(c, 2)] + [pq[m, x] for x in c]
        res.append((ms, ps, c, m, pairs))
    res.sort()
    print("launch ms  frac | pass ms  frac | planes | masked | pairings plane-plane ..., masked-plane ... (GB/s / 10) | min  mean  #below-6.45")
    for ms, ps, c, m, pairs in res:
        print("%8.4f %.3f | %7.4f %.3f | %s | m%d | %s | %4.0f %4.0f %d" % (ms, alg / ms * 1e-9 / 8, ps, alg / ps * 1e-9 / 8, " ".join("%d" % x for x in c), m,
                                                                    " ".join("%4.0f" % (x / 10) for x in pairs), min(pairs) / 10, statistics.mean(pairs) / 10, sum(1 for x in pairs if x < 6450)))
    sys.stdout.flush()


if __name__ == "__main__":
    main()
