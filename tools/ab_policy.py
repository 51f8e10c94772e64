#!/usr/bin/env python3
"""In-process A/B of the masked rows' store policy (hrx_kernel_pm.hip octets_out, hrx_kernel_pmd.hip): the SAME buffers, the same placement, modes alternating launch
block by launch block.  Needs the ntenv build (make -C halo2_regex_amd/csrc ntenv: the release kernels, HRX_NT_MIX / HRX_NT_FLAGS read at every launch):
    HRX_LIB_PATH=halo2_regex_amd/csrc/libhrx_ntenv.so python3 tools/ab_policy.py --config dfa256 --batch 131072 --len 4095 --rows 4096
modes (kNtMix* of csrc/hrx_kernel.hpp): streamed (0x200 = kNtMixNoOpenSpan: every masked row non-temporal, round 3's behaviour), open-span (0: the shipped rule — the masked
rows of a tile into which an open optimistic span reaches are written back), all-wb (0x100 = kNtMixMaskedWb: every masked row written back)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

def main():
    args = bench.parse_args(sys.argv[1:])
    import numpy as np, torch
    import halo2_regex_amd as hra
    assert any(k in hra.LIB_PATH for k in ("ntenv", "ablation", "stamps")), "HRX_LIB_PATH must name libhrx_ntenv.so (make ntenv: the release kernels, the store-policy variables read per launch)"
    dev = torch.device("cuda", 0)
    names, label, alphabet, gen, planted = bench.workload(args)
    M, n, B = args.rows, args.n, args.batch
    stride = (max(n, 1) + 15) // 16 * 16
    defs = [hra.RegexDefs(hra.AllstrRegexDef(a), [hra.SubstrRegexDef(t) for t in subs]) for a, subs in names]
    cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
    chars, lens = gen(B, n, seed=0, stride=stride)
    d_lens = torch.from_numpy(lens.astype(np.int32)).to(dev)
    d_chars = torch.from_numpy(chars).to(dev)
    D = len(names)
    foot = B * stride + B * M * 4 * D + B * M * 2
    nsets = max(2, min(8, int((48 << 30) // foot)))
    shift = (B // nsets + 37) % B
    sets = []
    for k in range(nsets):
        c = d_chars if k == 0 else torch.roll(d_chars, shifts=k * shift, dims=0)
        l = d_lens if k == 0 else torch.roll(d_lens, shifts=k * shift, dims=0)
        sets.append((hra.chars_to_position_major(c), l, cfg.alloc_outputs_position_major(B, dev)))
    torch.cuda.synchronize()
    launch = lambda i: cfg.witness_batch_position_major(sets[i % nsets][0], sets[i % nsets][1], out=sets[i % nsets][2], chars_pm_stride=stride)
    K = args.steps
    modes = [("streamed", "0x200"), ("open-span", "0"), ("all-wb", "0x100")]
    if os.environ.get("AB_MODES"):      # name=flags,name=flags (0x100: every masked row written back; 0x800: those of odd tiles)
        modes = [tuple(x.split("=")) for x in os.environ["AB_MODES"].split(",")]
    res = {m: [] for m, _ in modes}
    for rnd in range(5):
        for m, flag in modes:
            if ":" in flag:      # mix:flags — HRX_NT_MIX replaces the planner's value (its low byte: every k-th tile's records written back)
                os.environ["HRX_NT_MIX"], os.environ["HRX_NT_FLAGS"] = flag.split(":")
            else:
                os.environ.pop("HRX_NT_MIX", None)
                os.environ["HRX_NT_FLAGS"] = flag
            for i in range(nsets):
                launch(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(K):
                launch(i)
            e1.record(); torch.cuda.synchronize()
            res[m].append(e0.elapsed_time(e1) / K)
    rows = int(lens.sum())
    bpr = 4 * D + 3
    print("%s B=%d M=%d sets=%d K=%d (eager launches, events)" % (args.config, B, M, nsets, K))
    for m, _ in modes:
        v = sorted(res[m][1:])
        med = v[len(v) // 2]
        print("  %-10s ms/launch %s  median %.4f  frac %.3f" % (m, " ".join("%.4f" % x for x in res[m]), med, rows * bpr / (med * 1e-3) / 8e12))

if __name__ == "__main__":
    main()
