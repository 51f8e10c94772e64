#!/usr/bin/env python3
"""bench.py — DFA witness rows/sec on BASELINE.json configs[1]:
regex1_test DFA (+ substr1), batch 65536 x 1024-byte strings per MI355X.

  python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (one hrx_witness_batch_device_layout launch) over one device-resident batch.
Strings shard by index across ranks with no data-path collective (weak scaling: 65536 strings per GPU).

N > 1 runs one process per GPU either way:
  * under a launcher (torch.distributed.run sets RANK / WORLD_SIZE / MASTER_*): torch.distributed (backend nccl = RCCL)
    carries the barrier and the gather of the per-rank results, nothing else;
  * as a bare command: this process touches no GPU; it spawns N fresh children (one per device, HRX_BENCH_DEVICES=0,1,..
    overrides the rank -> device map), they rendezvous over gloo on 127.0.0.1 for the barrier and hand their
    (rows, elapsed) back over a pipe — the data path runs without RCCL.
Rank 0 (or the parent) prints ONE JSON line: `value` = rows of all ranks / max-over-ranks elapsed of the K timed steps.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_ROW = lambda D: 1 + 4 * D + 2   # 1 B char read, 4 B record per def + 2 B masked written
HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: 8.0 TB/s spec
RESULT_TAG = "HRX_BENCH_RANK_RESULT "


def parse_args(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=65536, help="strings per GPU")
    ap.add_argument("--len", type=int, default=1023, dest="n", help="bytes per string (n); rows M = --rows")
    ap.add_argument("--rows", type=int, default=1024, help="max_chars_size M (witness rows per string)")
    ap.add_argument("--dist", choices=["planted", "noise"], default="planted")
    ap.add_argument("--config", choices=["regex1", "regex23", "regex123", "headers3", "headers5", "dfa256"], default="regex1",
                    help="regex1: BASELINE configs[1] (the metric's workload); regex23: configs[2] shape (D=2); regex123: D=3 with the "
                    "reference's three DFAs; headers3: configs[3] shape (D=3 from/to/subject header definitions, 5 substrs); dfa256: configs[4] shape (synthetic total 256-state DFA over all 256 byte values)")
    ap.add_argument("--substr-pairs", type=int, default=200, help="dfa256: transitions in the random substring definition")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the comparison of the timed buffers with the oracle")
    ap.add_argument("--no-spread", action="store_true", help="skip the extra replays that give the per-step spread")
    ap.add_argument("--eager", action="store_true", help="launch the timed steps one by one instead of replaying a HIP graph of them")
    ap.add_argument("--dense", action="store_true", help="string-major: power-of-two pitches (M rows per string, n rounded to 16 "
                    "bytes) instead of hrx_recommended_pitches")
    ap.add_argument("--layout", choices=["position-major", "string-major"], default="position-major",
                    help="buffer layout of include/hrx.h: HRX_LAYOUT_POSITION_MAJOR (input and outputs chunked [pos/k][string][k], "
                    "the coalesced layout) or HRX_LAYOUT_STRING_MAJOR")
    ap.add_argument("--allow-debug-flags", action="store_true", help="tools only: run although HRX_DEBUG_FLAGS is set (recorded in the line's debug_flags)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc child passes that measure the launch's HBM traffic (roofline.traffic)")
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)   # spawned by a bare --gpus N run
    args = ap.parse_args(argv)
    args.argv = [a for a in argv if a != "--child"]
    return args


# ----------------------------------------------------------------------------------------------------------------
# checker / baseline legs (the oracle is never on the product path)
# ----------------------------------------------------------------------------------------------------------------
def oracle_handle(names):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import OracleDefs, load_oracle
    return OracleDefs(load_oracle(), names)


def cpu_baseline(o, names, chars, lens, M, budget_s=8.0):
    """The oracle (oracle/hrx_oracle.c, the reference-faithful C port) timed single-threaded on a bounded sample of
    the same workload — the reference itself is single-threaded — plus, as extra context, the same port
    over all host cores and the oracle's dense-table "best CPU" variant.  Returns (json object, oracle outputs of the sample)."""
    import numpy as np
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


def verify_timed_buffers(o, hra, out, chars, lens, M, D, pm, nstr):
    """Bit-exact comparison of the buffers the TIMED launches wrote with the oracle's rows for the first nstr strings
    (all host cores): the bench line then carries proof that the timed kernel did the work."""
    import numpy as np
    cores = os.cpu_count() or 1
    orec, omsk, ost = o.witness_batch(chars[:nstr], lens[:nstr], M, threads=cores)
    rec, msk, st = out
    if pm:
        # the first block of the position-major buffers holds strings 0 .. min(B, 65536) - 1
        nb = min(len(lens), hra.PM_BLOCK)
        q4, q8 = (M + 3) // 4, (M + 7) // 8
        r = rec[:q4 * D * nb * 4].reshape(q4, D, nb, 4)[:, :, :nstr].permute(2, 0, 3, 1).reshape(nstr, -1, D)[:, :M]
        m = msk[:q8 * nb * 8].reshape(q8, nb, 8)[:, :nstr].permute(1, 0, 2).reshape(nstr, -1)[:, :M]
    else:
        r, m = rec[:nstr], msk[:nstr]
    g_rec = r.cpu().numpy().view(np.uint32)
    g_msk = m.cpu().numpy().view(np.uint16)
    g_st = st[:nstr].cpu().numpy().view(np.uint64)
    ok = (ost & np.uint64(0xff)) == 0
    exact = bool(np.array_equal(g_st, ost) and np.array_equal(g_rec[ok], orec[ok]) and np.array_equal(g_msk[ok], omsk[ok]))
    return {"strings": int(nstr), "rows": int(lens[:nstr].sum()), "bit_exact": exact,
            "against": "oracle/hrx_oracle.c (%d threads) on the first %d strings of the timed output buffers: status words, records and masked rows" % (cores, nstr)}


def mix_ceiling(dev_index):
    """No-compute ceiling of the bench line's traffic mix on THIS box (tools/mixceil, built by __graft_entry__.build()):
    the kernel's own address streams (64 MiB read, 384 MiB written in position-major slabs) issued by 4 reader + 4 writer
    waves per CU with nothing else to do, and a plain dwordx4 copy of the same byte count.  Runs after the timed region."""
    exe = os.path.join(ROOT, "tools", "mixceil")
    if not os.path.exists(exe) or under_profiler():
        return None
    try:
        env = dict(os.environ, HIP_VISIBLE_DEVICES=str(dev_index))
        txt = subprocess.run([exe, "--brief"], capture_output=True, text=True, timeout=120, env=env).stdout
        res = {}
        for line in txt.splitlines():
            if line.startswith("MIXCEIL "):
                _, key, us = line.split()
                res[key] = float(us)
        return res or None
    except Exception as e:                                   # a probe must never break the bench line
        sys.stderr.write("mixceil failed: %s\n" % e)
        return None


def under_profiler():
    """True when this process itself runs under rocprofv3 (its tool library is preloaded): no nested profiler runs then, and no
    other helper processes either — a process forked from a profiled one must not exec."""
    return any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_CTOR"))


def measured_traffic(argv, dev_index):
    """HBM bytes per launch of THIS workload on THIS box: two short child runs of this script under `rocprofv3 --pmc`
    (FETCH_SIZE and WRITE_SIZE in separate passes, eager launches, nothing else), counters corrected as
    MI355X_MICROARCH.md prescribes (both in KiB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 B: x 2).  The children are
    separate processes started after the timed region; None if rocprofv3 is missing or a pass fails."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe or under_profiler():
        return None
    keep = [a for a in argv if a not in ("--no-spread", "--no-verify", "--no-cpu-baseline", "--eager")]
    for flag in ("--steps", "--warmup", "--gpus"):          # the child runs 3 eager launches on one device
        while flag in keep:
            i = keep.index(flag)
            del keep[i:i + 2]
    out = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            tmp = tempfile.mkdtemp(prefix="hrx_pmc_", dir="/tmp")
            env = dict(os.environ, TMPDIR="/tmp", HIP_VISIBLE_DEVICES=str(dev_index))
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", tmp, "-o", "r1", "--", sys.executable, os.path.abspath(__file__)] + keep + \
                  ["--eager", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-verify", "--no-spread", "--no-pmc"]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd="/tmp")
            vals = []
            for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "witness" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                        vals.append(float(row["Counter_Value"]))
            shutil.rmtree(tmp, ignore_errors=True)
            if r.returncode != 0 or not vals:
                sys.stderr.write("rocprofv3 --pmc %s pass failed (rc %d): traffic falls back to the committed profile\n" % (counter, r.returncode))
                return None
            out[counter] = sum(vals) / len(vals)
    except Exception as e:                                   # a probe must never break the bench line
        sys.stderr.write("pmc passes failed: %s\n" % e)
        return None
    read, written = out["FETCH_SIZE"] * 1024 * 2, out["WRITE_SIZE"] * 1024
    return {"read": read, "written": written, "total": read + written,
            "how": "two rocprofv3 --pmc child passes of this command on this box after the timed region (FETCH_SIZE x 2 KiB: gfx950; WRITE_SIZE KiB), means over the witness kernel's launches"}


# ----------------------------------------------------------------------------------------------------------------
# one rank
# ----------------------------------------------------------------------------------------------------------------
def workload(args):
    import numpy as np
    import halo2_regex_amd as hra
    from halo2_regex_amd import synth
    DFA_DIR = os.path.join(ROOT, "tests", "golden", "dfa")
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
    elif args.config == "headers5":     # D = 5: more than one launch walks side by side -> passes over groups of defs + combine
        hdr = lambda n, ns: (rd(n + "_lookup.txt"), [rd("%s_substr%d.txt" % (n, k)) for k in range(ns)])
        names, label = [hdr("header_from", 1), hdr("header_to", 1), hdr("header_subject", 3), pair(1), pair(2)], "from/to/subject header definitions + regex1 + regex2 (7 substrs, multi-pass)"
        gen = synth.headers_planted if args.dist == "planted" else synth.noise
    else:
        allb = np.arange(256, dtype=np.uint8)
        a_txt, sub_txt = synth.random_dfa(256, seed=2, alphabet=allb, n_substr_pairs=args.substr_pairs)
        names, label, alphabet = [(a_txt.encode(), [sub_txt.encode()])], "synthetic total DFA 256 states x 256 symbols (seed 2)", "all 256 byte values"
        gen = lambda B, n, seed=0, stride=None: synth.noise(B, n, seed=seed, alphabet=allb, stride=stride)
    planted = gen in (synth.regex1_planted, synth.regex23_planted, synth.headers_planted)
    return names, label, alphabet, gen, planted


def run_rank(args, rank, world, device_index, barrier):
    """Everything one rank does; returns its result dict (rank 0's carries the line's descriptive fields)."""
    import numpy as np
    import torch
    import halo2_regex_amd as hra

    dbg = os.environ.get("HRX_DEBUG_FLAGS", "")
    if dbg.strip() not in ("", "0", "0x0") and not args.allow_debug_flags:
        raise SystemExit("bench.py refuses to run with HRX_DEBUG_FLAGS=%r set: the timed kernel must be the one the planner picks" % dbg)
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    names, label, alphabet, gen, planted = workload(args)
    D = len(names)
    M, n, B = args.rows, args.n, args.batch
    pm = args.layout == "position-major"
    rec_pitch, msk_pitch, rec_stride = hra.recommended_pitches(M)
    if args.dense or pm:
        rec_pitch, msk_pitch, rec_stride = M, M, (max(n, 1) + 15) // 16 * 16
    stride = rec_stride
    defs = [hra.RegexDefs(hra.AllstrRegexDef(a), [hra.SubstrRegexDef(t) for t in subs]) for a, subs in names]
    cfg = hra.RegexVerifyConfig.configure(M, defs, device=device_index)

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

    def sync_barrier():
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()

    # the kernel and geometry the planner picks for this shape on this device (what rocprofv3 will list)
    desc = cfg.describe_launch(B, layout=3 if pm else 0, num_cus=torch.cuda.get_device_properties(dev).multi_processor_count)

    for _ in range(args.warmup):
        step()
    sync_barrier()
    # The K timed steps are K kernel launches.  They are recorded once into a HIP graph (stream capture of the very same
    # step() calls) and the graph is replayed inside the timed region, so that a slow host thread cannot turn the
    # measurement into a launch-rate test (one launch is ~80 us of device time); --eager launches them one by one.
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
    # poison the outputs: what the verification reads afterwards was written by the timed launches
    for t in out[:2]:
        t.fill_(-1)
    sync_barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run_steps()                                            # EXACTLY K steps
    ev1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    kern_ms = ev0.elapsed_time(ev1) / args.steps      # average launch duration, HIP events on the launch stream

    # sanity: every string of the timed workload finished with status 0
    st = out[2].cpu().numpy().view(np.uint64)
    assert ((st & np.uint64(0xff)) == 0).all(), "status != ok in the bench workload"

    res = {"rank": rank, "device": device_index, "rows": rows_per_step * args.steps, "elapsed_s": elapsed, "avg_launch_ms": kern_ms,
           "debug_flags": dbg or None}
    if rank != 0:
        return res

    o = None
    if not args.no_verify or not args.no_cpu_baseline:
        o = oracle_handle(names)
    if not args.no_verify:
        res["verified"] = verify_timed_buffers(o, hra, out, chars, lens, M, D, pm, min(B, 16384))
        if not res["verified"]["bit_exact"]:
            raise SystemExit("bench.py: the timed output buffers differ from the oracle")
    # spread: R more replays of the same K steps, each timed by its own event pair (after the contract's timed region)
    if not args.no_spread:
        per = []
        reps = max(5, min(200, int(0.25 / max(kern_ms * args.steps * 1e-3, 1e-6))))
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run_steps(); e1.record()
            torch.cuda.synchronize()
            per.append(e0.elapsed_time(e1) / args.steps)
        res["spread"] = {"replays": reps, "steps_per_replay": args.steps, "ms_per_step_median": statistics.median(per),
                         "ms_per_step_min": min(per), "ms_per_step_max": max(per)}
    # The timed steps re-process ONE batch into ONE set of output buffers (the contract's step), so from the second launch on the
    # 256-MB Infinity Cache holds part of what a launch reads and overwrites.  The complementary figure: the same launches over
    # NSETS input / output sets used in turn — nothing a launch touches was touched by the previous NSETS - 1 launches.
    if not args.no_spread and world == 1 and pm:
        try:
            nsets = 8
            foot = B * stride + sum(t.numel() * t.element_size() for t in out)
            if nsets * foot <= (24 << 30):
                sets = [(d_chars, out)]
                for k in range(1, nsets):
                    c2, _ = gen(B, n, seed=1000 + k, stride=stride)
                    sets.append((hra.chars_to_position_major(torch.from_numpy(c2).to(dev)), cfg.alloc_outputs_position_major(B, dev)))
                rot = lambda i: cfg.witness_batch_position_major(sets[i % nsets][0], d_lens, out=sets[i % nsets][1], chars_pm_stride=stride)
                for i in range(2 * nsets):
                    rot(i)
                torch.cuda.synchronize()
                kk = max(nsets, min(200, args.steps) // nsets * nsets)
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.stream(side):
                    with torch.cuda.graph(g2, stream=side, capture_error_mode="thread_local"):
                        for i in range(kk):
                            rot(i)
                torch.cuda.current_stream(dev).wait_stream(side)
                g2.replay()
                torch.cuda.synchronize()
                per = []
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); g2.replay(); e1.record()
                    torch.cuda.synchronize()
                    per.append(e0.elapsed_time(e1) / kk)
                res["fresh_buffers"] = {"sets": nsets, "ms_per_step_median": statistics.median(per), "ms_per_step_min": min(per), "ms_per_step_max": max(per)}
                del sets, g2
        except Exception as e:                                   # a probe must never break the bench line
            sys.stderr.write("fresh-buffer probe failed: %s\n" % e)
    res["desc"] = desc
    res["config"] = {"workload": "%s DFA (D=%d), %d x %d-byte strings per GPU (n=%d chars, M=%d witness rows), %s"
                                 % (label, D, B, stride, n, M, "uniform noise over the %s%s" % (alphabet, " + planted match" if planted else "")),
                     "batch_per_gpu": B, "n": n, "max_chars_size": M, "defs": D, "rows_counted": "sum of n (character positions)",
                     "buffers": ("HRX_LAYOUT_POSITION_MAJOR | HRX_LAYOUT_INPUT_POSITION_MAJOR (blocks of 65536 strings): chars [%d/16][B][16], records "
                                 "[M/4][D][B][4], masked [M/8][B][8] (include/hrx.h); outputs from hrx_alloc_outputs_position_major (placement-aware "
                                 "from 1 GiB of records on, two plain allocations below)" % stride) if pm else
                                ("string-major; input stride %d B, records pitch %d rows, masked pitch %d rows" % (stride, rec_pitch, msk_pitch)),
                     "sharding": "by string index, no collective", "launch_mode": launch_mode}
    res["D"], res["rows_per_step"] = D, rows_per_step
    if world == 1:
        del d_chars
        torch.cuda.empty_cache()
        if args.config == "regex1" and B == 65536 and M == 1024 and pm:
            res["mix_ceiling"] = mix_ceiling(device_index)
        if not args.no_pmc:
            res["traffic"] = measured_traffic(args.argv, device_index)
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(o, names, chars, lens, M)
    return res


# ----------------------------------------------------------------------------------------------------------------
# aggregation (imported by tests/test_dist_cpu.py)
# ----------------------------------------------------------------------------------------------------------------
def pmc_traffic(args):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (profiles/*_pmc.json:
    WRITE_SIZE + 2 x FETCH_SIZE, the gfx950 correction of MI355X_MICROARCH.md).  bench.py cannot collect counters
    itself; None unless the committed passes were taken on the workload and kernel being run."""
    try:
        p = json.load(open(os.path.join(ROOT, "profiles", "r02_pm_pmc.json" if args.layout == "position-major" else "r01_split_pmc.json")))
        if (args.config == "regex1" and args.batch == 65536 and args.n == 1023 and args.rows == 1024 and args.dist == "planted"
                and not args.dense):
            return p["hbm_bytes_per_launch"]["total"]
    except Exception:
        pass
    return None


def aggregate(per_rank, args):
    """The JSON line from the per-rank results: value = rows of ALL ranks / MAX over ranks of the elapsed time."""
    per_rank = sorted(per_rank, key=lambda r: r["rank"])
    world = len(per_rank)
    r0 = per_rank[0]
    elapsed = max(r["elapsed_s"] for r in per_rank)
    total_rows = sum(r["rows"] for r in per_rank)
    value = total_rows / elapsed
    D = r0["D"]
    algo_bytes = BYTES_PER_ROW(D) * r0["rows_per_step"]
    kern_ms = r0["avg_launch_ms"]
    achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
    line = {
        "metric": "DFA witness rows/sec", "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": r0["config"],
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": (r0.get("traffic") or {}).get("total") or pmc_traffic(args),
                     "kernel": r0["desc"].split(" grid=")[0], "launch": "grid=" + r0["desc"].split(" grid=")[1],
                     "avg_launch_ms": kern_ms,
                     "algorithmic_bytes_per_launch": algo_bytes, "bytes_per_row": BYTES_PER_ROW(D)},
        "ranks_seen": world,
        "per_rank": [{"rank": r["rank"], "device": r["device"], "rows": r["rows"], "elapsed_s": r["elapsed_s"],
                      "rows_per_s": r["rows"] / r["elapsed_s"], "avg_launch_ms": r["avg_launch_ms"]} for r in per_rank],
        "debug_flags": r0.get("debug_flags"),
    }
    line["roofline"]["traffic_source"] = r0["traffic"] if r0.get("traffic") else ("profiles/r02_pm_pmc.json (committed rocprofv3 PMC passes of this command)" if line["roofline"]["traffic"] else None)
    if r0.get("verified"):
        line["verified"] = r0["verified"]
    if r0.get("spread"):
        line["spread"] = r0["spread"]
    if r0.get("fresh_buffers"):
        fb = r0["fresh_buffers"]
        fb_gbs = algo_bytes / (fb["ms_per_step_median"] * 1e-3) / 1e9
        line["roofline"]["fresh_buffers"] = dict(fb, achieved=fb_gbs, frac=fb_gbs / HBM_PEAK_GBS,
                                                 what="the same launches over %d input / output buffer sets used in turn (nothing a launch touches is left in the 256-MB "
                                                      "Infinity Cache by the previous launches); the timed steps above re-process one batch into one set of buffers" % fb["sets"])
    mc = r0.get("mix_ceiling")
    if mc:
        # the kernel's traffic mix with no compute, measured on this box after the timed region (tools/mixceil.cpp)
        same = {k: v for k, v in mc.items() if not k.endswith("_fresh")}
        fresh = {k: v for k, v in mc.items() if k.endswith("_fresh")}
        best = min(same.values())
        line["roofline"]["mix_ceiling"] = {"us_per_launch": same, "best_us": best, "best_gbs": algo_bytes / (best * 1e-6) / 1e9,
                                           "kernel_over_best": kern_ms * 1e3 / best,
                                           "what": "tools/mixceil --brief: the same 64 MiB read + 384 MiB written per launch, no DFA work: "
                                                   "copy = plain dwordx4 copy of the byte count; pair / pair_nt / pair_mix = the kernel's position-major slabs "
                                                   "from 4 reader + 4 writer waves per CU with write-back / streaming stores / the shipped mix (streaming, every other "
                                                   "tile's records write-back); like the timed steps, these probes re-write one set of buffers"}
        if fresh:
            fbest = min(fresh.values())
            line["roofline"]["mix_ceiling"]["fresh_buffers"] = {
                "us_per_launch": fresh, "best_us": fbest, "best_gbs": algo_bytes / (fbest * 1e-6) / 1e9,
                "what": "the pair probes over 8 buffer sets used in turn: what HBM alone sustains for this traffic mix"}
            if line["roofline"].get("fresh_buffers"):
                line["roofline"]["fresh_buffers"]["kernel_over_best_probe"] = line["roofline"]["fresh_buffers"]["ms_per_step_median"] * 1e3 / fbest
    if r0.get("cpu_baseline"):
        line["cpu_baseline"] = r0["cpu_baseline"]
    return line


def gather_and_aggregate(res, args):
    """Under torch.distributed: gather every rank's result dict on all ranks; rank 0 returns the line, the others None."""
    import torch.distributed as dist
    world = dist.get_world_size()
    allres = [None] * world
    dist.all_gather_object(allres, res)
    return aggregate(allres, args) if dist.get_rank() == 0 else None


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_children(args, argv):
    """Bare `python bench.py --gpus N`: N fresh child processes, one per device.  This (parent) process never touches a GPU."""
    devs = os.environ.get("HRX_BENCH_DEVICES")
    devices = [int(x) for x in devs.split(",")] if devs else list(range(args.gpus))
    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, HRX_BENCH_RANK=str(r), HRX_BENCH_WORLD=str(args.gpus), HRX_BENCH_PORT=str(port),
                   HRX_BENCH_DEVICE=str(devices[r % len(devices)]))
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv + ["--child"], env=env, stdout=subprocess.PIPE, text=True))
    results, failed = [], False
    for r, p in enumerate(procs):
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write("bench.py: rank %d exited with %d\n" % (r, p.returncode))
            failed = True
            continue
        lines = [l for l in out.splitlines() if l.startswith(RESULT_TAG)]
        if not lines:
            sys.stderr.write("bench.py: rank %d reported nothing\n" % r)
            failed = True
            continue
        results.append(json.loads(lines[-1][len(RESULT_TAG):]))
    if failed or len(results) != args.gpus:
        raise SystemExit(1)
    return aggregate(results, args)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if "RANK" in os.environ and "MASTER_ADDR" in os.environ and not args.child:
        # ---- launched by torch.distributed.run: one rank per GPU, RCCL for the barrier and the gather only
        import torch
        import torch.distributed as dist
        world, rank, local_rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
        if world != args.gpus:
            raise SystemExit("--gpus %d but the launcher started %d ranks" % (args.gpus, world))
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        res = run_rank(args, rank, world, local_rank, dist.barrier)
        line = gather_and_aggregate(res, args)
        if line is not None:
            print(json.dumps(line), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    if args.child:
        # ---- one of the N children of a bare --gpus N run: gloo over 127.0.0.1 for the barrier, result over the pipe
        import torch.distributed as dist
        rank, world = int(os.environ["HRX_BENCH_RANK"]), int(os.environ["HRX_BENCH_WORLD"])
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["HRX_BENCH_PORT"], rank=rank, world_size=world)
        res = run_rank(args, rank, world, int(os.environ["HRX_BENCH_DEVICE"]), dist.barrier)
        print(RESULT_TAG + json.dumps(res), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    if args.gpus > 1:
        print(json.dumps(spawn_children(args, argv)), flush=True)
        return
    res = run_rank(args, 0, 1, 0, lambda: None)
    print(json.dumps(aggregate([res], args)), flush=True)


if __name__ == "__main__":
    main()
