#!/usr/bin/env python3
"""bench.py — DFA witness rows/sec on BASELINE.json configs[1]:
regex1_test DFA (+ substr1), batch 65536 x 1024-byte strings per MI355X.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path (one hrx_witness_batch_device launch) over one device-resident batch.
Strings shard by index across ranks with no data-path collective (weak scaling: 65536 strings per GPU);
torch.distributed (RCCL) is used only for the barrier and the max-over-ranks of the elapsed time.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_ROW = lambda D: 1 + 4 * D + 2   # 1 B char read, 4 B record per def + 2 B masked written
HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(names, chars, lens, M, budget_s=8.0):
    """The oracle (oracle/hrx_oracle.c, the reference-faithful C port) timed single-threaded on a bounded sample of
    the same workload — the reference itself is single-threaded — plus, as extra context, the same port
    over all host cores and the oracle's dense-table "best CPU" variant.  Checker/baseline only — never on the
    product path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from oracle_lib import OracleDefs, load_oracle
    o = OracleDefs(load_oracle(), names)
    nstr = min(len(chars), 16384)
    cores = os.cpu_count() or 1
    ndefs = len(names)
    out = np.zeros((nstr, M, ndefs), np.uint32), np.zeros((nstr, M), np.uint16), np.zeros(nstr, np.uint64)
    rows1 = int(lens[:nstr].sum())

    def timed(budget, **kw):
        t0 = time.perf_counter()
        o.witness_batch(chars[:nstr], lens[:nstr], M, out=out, **kw)
        dt1 = max(time.perf_counter() - t0, 1e-6)
        reps = max(1, int(budget / dt1))
        t0 = time.perf_counter()
        for _ in range(reps):
            o.witness_batch(chars[:nstr], lens[:nstr], M, out=out, **kw)
        dt = time.perf_counter() - t0
        return rows1 * reps / dt, reps, dt

    v, reps, dt = timed(budget_s)
    res = {"value": v, "unit": "rows/s", "cores": 1, "kind": "port",
           "sample": "first %d strings of the same batch x %d passes (%d rows), oracle/hrx_oracle.c -O3, 1 thread, %.1f s; "
                     "host has %d cores" % (nstr, reps, rows1 * reps, dt, cores)}
    v, reps, dt = timed(3.0, threads=cores)
    res["all_cores"] = {"value": v, "unit": "rows/s", "cores": cores, "kind": "port",
                        "sample": "same sample x %d passes, one string per task" % reps}
    v, reps, dt = timed(2.0, dense=True)
    res["dense_table"] = {"value": v, "unit": "rows/s", "cores": 1,
                          "kind": "dense-table CPU variant of the oracle (not the reference's data structures)",
                          "sample": "same sample x %d passes" % reps}
    v, reps, dt = timed(2.0, dense=True, threads=cores)
    res["dense_table_all_cores"] = {"value": v, "unit": "rows/s", "cores": cores, "kind": "dense-table CPU variant of the oracle",
                                    "sample": "same sample x %d passes" % reps}
    return res


def copy_ceiling_gbs(dev, nbytes):
    """Measured device-copy ceiling of this box: a plain torch copy moving the same number of bytes as one
    launch (half read, half written).  Context for roofline.frac, which stays priced against the 8 TB/s spec."""
    import torch
    x = torch.empty(nbytes // 2, dtype=torch.uint8, device=dev)
    y = torch.empty_like(x)
    for _ in range(3):
        y.copy_(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y.copy_(x)
    e1.record()
    torch.cuda.synchronize()
    return 2 * x.numel() * 20 / (e0.elapsed_time(e1) * 1e-3) / 1e9


def pmc_traffic(args, D):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (profiles/*_pmc.json:
    WRITE_SIZE + 2 x FETCH_SIZE, the gfx950 correction of MI355X_MICROARCH.md).  bench.py cannot collect counters
    itself; None unless the committed passes were taken on the workload being run."""
    try:
        p = json.load(open(os.path.join(ROOT, "profiles", "r01_pm_pmc.json" if args.layout == "position-major" else "r01_split_pmc.json")))
        if (args.config == "regex1" and args.batch == 65536 and args.n == 1023 and args.rows == 1024 and args.dist == "planted"
                and not args.dense):
            return p["hbm_bytes_per_launch"]["total"]
    except Exception:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=65536, help="strings per GPU")
    ap.add_argument("--len", type=int, default=1023, dest="n", help="bytes per string (n); rows M = --rows")
    ap.add_argument("--rows", type=int, default=1024, help="max_chars_size M (witness rows per string)")
    ap.add_argument("--dist", choices=["planted", "noise"], default="planted")
    ap.add_argument("--config", choices=["regex1", "regex23", "regex123", "headers3", "dfa256"], default="regex1",
                    help="regex1: BASELINE configs[1] (the metric's workload); regex23: configs[2] shape (D=2); regex123: D=3 with the "
                    "reference's three DFAs; headers3: configs[3] shape (D=3 from/to/subject header definitions, 5 substrs); dfa256: configs[4] shape (synthetic total 256-state DFA over all 256 byte values)")
    ap.add_argument("--substr-pairs", type=int, default=200, help="dfa256: transitions in the random substring definition (sets how often "
                    "start / end events occur and how long optimistic reveal spans run before they are confirmed or repaired; DESIGN.md §4.2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="launch the timed steps one by one instead of replaying a HIP graph of them")
    ap.add_argument("--dense", action="store_true", help="string-major: power-of-two pitches (M rows per string, n rounded to 16 "
                    "bytes) instead of hrx_recommended_pitches")
    ap.add_argument("--layout", choices=["position-major", "string-major"], default="position-major",
                    help="buffer layout of include/hrx.h: HRX_LAYOUT_POSITION_MAJOR (input and outputs chunked [pos/k][string][k], "
                    "the coalesced layout) or HRX_LAYOUT_STRING_MAJOR")
    args = ap.parse_args()

    import numpy as np
    import torch
    import halo2_regex_amd as hra
    from halo2_regex_amd import synth
    DFA_DIR = os.path.join(ROOT, "tests", "golden", "dfa")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ   # launched by torch.distributed.run
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    rd = lambda f: open(os.path.join(DFA_DIR, f), "rb").read()
    pair = lambda k: (rd("regex%d_test_lookup.txt" % k), [rd("substr%d_test_lookup.txt" % k)])
    alphabet = "98-byte alphabet"
    if args.config == "regex1":
        names, label = [pair(1)], "regex1_test+substr1"
        gen = synth.regex1_planted if args.dist == "planted" else synth.noise
    elif args.config == "regex23":
        names, label = [pair(2), pair(3)], "regex2+regex3 with substrs"
        gen = synth.regex23_planted if args.dist == "planted" else synth.noise
    elif args.config == "regex123":
        names, label = [pair(1), pair(2), pair(3)], "regex1+regex2+regex3 with substrs"
        gen = synth.noise
    elif args.config == "headers3":
        hdr = lambda n, ns: (rd(n + "_lookup.txt"), [rd("%s_substr%d.txt" % (n, k)) for k in range(ns)])
        names, label = [hdr("header_from", 1), hdr("header_to", 1), hdr("header_subject", 3)], "from/to/subject header definitions (5 substrs)"
        gen = synth.headers_planted if args.dist == "planted" else synth.noise
    else:
        allb = np.arange(256, dtype=np.uint8)
        a_txt, sub_txt = synth.random_dfa(256, seed=2, alphabet=allb, n_substr_pairs=args.substr_pairs)
        names, label, alphabet = [(a_txt.encode(), [sub_txt.encode()])], "synthetic total DFA 256 states x 256 symbols (seed 2)", "all 256 byte values"
        gen = lambda B, n, seed=0, stride=None: synth.noise(B, n, seed=seed, alphabet=allb, stride=stride)
    D = len(names)
    M, n, B = args.rows, args.n, args.batch
    pm = args.layout == "position-major"
    rec_pitch, msk_pitch, rec_stride = hra.recommended_pitches(M)
    if args.dense or pm:
        rec_pitch, msk_pitch, rec_stride = M, M, (max(n, 1) + 15) // 16 * 16
    stride = rec_stride
    defs = [hra.RegexDefs(hra.AllstrRegexDef(a), [hra.SubstrRegexDef(t) for t in subs]) for a, subs in names]
    cfg = hra.RegexVerifyConfig.configure(M, defs, device=local_rank)

    # this rank's shard of the (world * B)-string job: independent strings, seeded per rank
    chars, lens = gen(B, n, seed=rank, stride=stride)
    d_chars = torch.from_numpy(chars).to(dev)
    d_lens = torch.from_numpy(lens.astype(np.int32)).to(dev)
    rows_per_step = int(lens.sum())
    if pm:
        d_chars = hra.chars_to_position_major(d_chars)       # [stride/16][B][16]: done once, outside the timed region
        out = cfg.alloc_outputs_position_major(B, dev)
        step = lambda: cfg.witness_batch_position_major(d_chars, d_lens, out=out, chars_pm_stride=stride)
    else:
        out = cfg.alloc_outputs(B, dev, pitched=not args.dense)
        step = lambda: cfg.witness_batch(d_chars, d_lens, out=out)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # the kernel and geometry the planner picks for this shape on this device (what rocprofv3 will list)
    desc = cfg.describe_launch(B, layout=3 if pm else 0, num_cus=torch.cuda.get_device_properties(dev).multi_processor_count)

    for _ in range(args.warmup):
        step()
    barrier()
    # The K timed steps are K kernel launches.  They are recorded once into a HIP graph (stream capture of the very same
    # step() calls) and the graph is replayed inside the timed region, so that a slow host thread cannot turn the
    # measurement into a launch-rate test (one launch is ~90 us of device time); --eager launches them one by one.
    run_steps, launch_mode = None, "eager"
    if not args.eager:
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                # thread_local: the RCCL watchdog thread of a multi-GPU run may poll events while this thread captures
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    for _ in range(args.steps):
                        step()
            torch.cuda.current_stream(dev).wait_stream(side)
            for _ in range(2):                             # untimed replays: graph upload, caches, clocks
                g.replay()
            torch.cuda.synchronize()
            run_steps, launch_mode = g.replay, "hipGraph of %d kernel nodes" % args.steps
        except Exception as e:                              # capture unsupported: fall back to plain launches
            sys.stderr.write("graph capture failed (%s): eager launches\n" % e)
            torch.cuda.synchronize()
    if run_steps is None:
        def run_steps():
            for _ in range(args.steps):
                step()   # launched on torch's current stream, where the events sit
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run_steps()
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = ev0.elapsed_time(ev1) / args.steps      # average launch duration, HIP events on the launch stream

    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity: every string of the timed workload finished with status 0 and the result is reproducible
    st = out[2].cpu().numpy().view(np.uint64)
    assert ((st & np.uint64(0xff)) == 0).all(), "status != ok in the bench workload"

    if rank == 0:
        total_rows = rows_per_step * world * args.steps
        value = total_rows / elapsed
        algo_bytes = BYTES_PER_ROW(D) * rows_per_step
        achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
        line = {
            "metric": "DFA witness rows/sec", "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s DFA (D=%d), %d x %d-byte strings per GPU (n=%d chars, M=%d witness rows), %s"
                                   % (label, D, B, stride, n, M, "uniform noise over the %s%s" % (
                                       alphabet, " + planted match" if gen in (synth.regex1_planted, synth.regex23_planted, synth.headers_planted) else "")),
                       "batch_per_gpu": B, "n": n, "max_chars_size": M, "defs": D, "rows_counted": "sum of n (character positions)",
                       "buffers": ("HRX_LAYOUT_POSITION_MAJOR | HRX_LAYOUT_INPUT_POSITION_MAJOR (blocks of 65536 strings): chars [%d/16][B][16], records "
                                   "[M/4][D][B][4], masked [M/8][B][8] (include/hrx.h)" % stride) if pm else
                                  ("string-major; input stride %d B, records pitch %d rows, masked pitch %d rows"
                                   % (stride, rec_pitch, msk_pitch)),
                       "sharding": "by string index, no collective", "launch_mode": launch_mode},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(args, D),
                         "kernel": desc.split(" grid=")[0], "launch": "grid=" + desc.split(" grid=")[1],
                         "avg_launch_ms": kern_ms,
                         "algorithmic_bytes_per_launch": algo_bytes, "bytes_per_row": BYTES_PER_ROW(D)},
        }
        if world == 1:
            ceil = copy_ceiling_gbs(dev, algo_bytes)
            line["roofline"]["measured_copy_ceiling"] = ceil
            line["roofline"]["frac_of_copy_ceiling"] = achieved / ceil
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(names, chars, lens, M)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
