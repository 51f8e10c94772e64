#!/usr/bin/env python3
"""Same process, same input, same lease: the interleaved position-major records ([M/4][D][B][4], one allocation + placed masked rows) against RECORD PLANES
(hrx_witness_batch_device_planes: every def's plane a buffer of its own, hrx_alloc_output_planes), and — def-parallel kernel, two and three defs — the last def's
walker combining against a combiner wave of its own (HRX_OPT_PMD_COMBINER_WAVE).  Every variant's rows are compared with variant 0's (which tests/ pin to the oracle).

  python tools/planes_ab.py --config headers3 --batch 32768 --len 32767 --rows 32768 --distinct 4096        # cfg 4 share
  python tools/planes_ab.py --config regex23 --batch 1048576 --len 2047 --rows 2048 --distinct 65536 --sets 1 # cfg 3 full
"""
import argparse
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
    ap.add_argument("--sets", type=int, default=2)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--fin", default="0,1", help="HRX_OPT_PMD_COMBINER_WAVE values to run (2 = off, 1 = on, 0 = default)")
    ap.add_argument("--no-compare", action="store_true")
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
    print("%s  D=%d  %d x %d B  rows/launch %d  algorithmic %.3f GB" % (label, D, B, stride, rows, alg * 1e-9), flush=True)
    fins = [int(x) for x in a.fin.split(",")]
    variants = []      # (name, cfg, launch(i), pass(i), outs)
    for planes in (False, True):
        for fin in fins:
            cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
            cfg.set_option(hra.OPT_PMD_COMBINER_WAVE, fin)
            outs, reps_ = [], []
            for _ in range(a.sets):
                outs.append(cfg.alloc_output_planes(B, dev, stripes=2 if D == 1 else None) if planes else cfg.alloc_outputs_position_major(B, dev))
                reps_.append(cfg.last_placement_report())
            if planes:
                launch = lambda i, cfg=cfg, outs=outs: cfg.witness_batch_planes(c_pm, d_lens, out=outs[i % len(outs)], chars_pm_stride=stride)
                tpass = lambda i, cfg=cfg, outs=outs: cfg.traffic_pass_planes(c_pm, B, outs[i % len(outs)], stride)
            else:
                launch = lambda i, cfg=cfg, outs=outs: cfg.witness_batch_position_major(c_pm, d_lens, out=outs[i % len(outs)], chars_pm_stride=stride)
                tpass = lambda i, cfg=cfg, outs=outs: cfg.traffic_pass(c_pm, B, outs[i % len(outs)], stride)
            name = "%s fin=%d" % ("planes     " if planes else "interleaved", fin)
            print("%s: %s | placement %s" % (name, cfg.describe_launch(B, layout=3 | (hra.LAYOUT_RECORD_PLANES if planes else 0)).split(" lds=")[0],
                                             ["steps %d ref %.0f best %.0f GB/s %.0f ms" % (r["steps"], r["ref_gbs"], r["best_gbs"], r["search_ms"]) for r in reps_]), flush=True)
            variants.append((name, cfg, launch, tpass, outs, planes))

    def timed(fn):
        per = []
        for _ in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(a.steps):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            per.append(e0.elapsed_time(e1) / a.steps)
        return per

    for name, cfg, launch, tpass, outs, planes in variants:      # warm-up + status
        for i in range(a.sets):
            launch(i)
        torch.cuda.synchronize()
        st = outs[0][2].cpu().numpy().view(np.uint64)
        assert ((st & np.uint64(0xff)) == 0).all(), name
    if not a.no_compare:
        ref = variants[0][4][0]
        q4 = (M + 3) // 4
        for name, cfg, launch, tpass, outs, planes in variants[1:]:
            same = bool(torch.equal(outs[0][1], ref[1])) and bool(torch.equal(outs[0][2], ref[2]))
            if planes and len(outs[0][0]) != D:      # one def in two row stripes
                a_r = hra.planes_to_string_major(outs[0][0], outs[0][1], B, M, D=D)[0]
                b_r = hra.position_major_to_string_major(ref[0], ref[1], B, M, D)[0]
                same = same and bool(torch.equal(a_r, b_r))
                print("%s: rows equal to variant 0's: %s" % (name, same), flush=True)
                assert same
                continue
            for blk in range(0, B, hra.PM_BLOCK):
                nb = min(hra.PM_BLOCK, B - blk)
                r0 = ref[0][blk * q4 * D * 4:][:q4 * D * nb * 4].view(q4, D, nb, 4)
                for d in range(D):
                    if planes:
                        same = same and bool(torch.equal(outs[0][0][d][blk * q4 * 4:][:q4 * nb * 4].view(q4, nb, 4), r0[:, d]))
                    else:
                        same = same and bool(torch.equal(outs[0][0][blk * q4 * D * 4:][:q4 * D * nb * 4].view(q4, D, nb, 4)[:, d], r0[:, d]))
            print("%s: rows equal to variant 0's: %s" % (name, same), flush=True)
            assert same
    res = {v[0]: [] for v in variants}
    for rnd in range(2):
        for name, cfg, launch, tpass, outs, planes in variants:
            res[name] += timed(launch)
    for name, per in res.items():
        ms = statistics.median(per)
        print("%-22s %.4f ms (min %.4f max %.4f)  %.3f TB/s  frac %.3f" % (name, ms, min(per), max(per), alg / ms * 1e-9, alg / ms * 1e-9 / 8), flush=True)
    for name, cfg, launch, tpass, outs, planes in variants:
        if name.endswith("fin=%d" % fins[0]):
            per = timed(tpass)
            ms = statistics.median(per)
            print("%-22s no-compute pass %.4f ms  %.3f TB/s  frac %.3f" % (name, ms, alg / ms * 1e-9, alg / ms * 1e-9 / 8), flush=True)


if __name__ == "__main__":
    main()
