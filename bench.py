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
    ap.add_argument("--config", choices=["regex1", "regex23", "regex123", "headers3", "headers4", "headers5", "dfa256"], default="regex1",
                    help="regex1: BASELINE configs[1] (the metric's workload); regex23: configs[2] shape (D=2); regex123: D=3 with the "
                    "reference's three DFAs; headers3: configs[3] shape (D=3 from/to/subject header definitions, 5 substrs); dfa256: configs[4] shape (synthetic total 256-state DFA over all 256 byte values)")
    ap.add_argument("--untimed-replays", type=int, default=0, help="untimed replays of the K-step graph before the timed one (default: ~100 ms of them)")
    ap.add_argument("--substr-pairs", type=int, default=200, help="dfa256: transitions in the random substring definition")
    ap.add_argument("--substr-defs", type=int, default=1, help="dfa256: substring definitions of the random DFA (SURVEY §8d cfg 5: 1-2), --substr-pairs tagged pairs each")
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
    ap.add_argument("--sets", type=int, default=8, help="input / output buffer sets the timed steps rotate over (1: every step re-processes one batch into one set of "
                    "buffers, which leaves part of the traffic in the 256-MB Infinity Cache)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="weak: --batch strings PER GPU; strong: --batch strings IN TOTAL, sharded by "
                    "string index over the ranks (hrx_shard_range) — the BASELINE multi-GPU configs: --config headers3 --batch 262144 --len 32768 --rows 32768, "
                    "--config dfa256 --batch 1048576 --len 4096 --rows 4096")
    ap.add_argument("--verify-all-ranks", action="store_true", help="every rank compares its own timed buffers with the oracle (default: rank 0)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc child passes that measure the launch's HBM traffic (roofline.traffic)")
    ap.add_argument("--distinct", type=int, default=0, help="generate this many DISTINCT strings and build the batch from rotated copies of them, block after block "
                    "(0 = every string of the batch distinct).  Planting text into 2^20 strings on the host takes minutes; the oracle then walks the distinct "
                    "strings once and EVERY string of every buffer set is compared with the rows of the string it is a copy of")
    ap.add_argument("--planes", action="store_true", help="position-major outputs as RECORD PLANES: every def's records in a buffer of its own "
                    "(hrx_witness_batch_device_planes, buffers from hrx_alloc_output_planes: a pool of candidates, their pairings, a dry launch); the line then also times the interleaved "
                    "layout / the pair walk's buffers over as many buffer sets (interleaved_layout).  One def: one records buffer from the same allocator (--stripes 2: two row stripes)")
    ap.add_argument("--stripes", type=int, default=1, choices=[1, 2], help="--planes with one def: the records in this many row stripes")
    ap.add_argument("--ctx-option", action="append", default=[], metavar="N=V", help="hrx_ctx_set_option(N, V) on the context before anything runs (include/hrx.h HRX_OPT_*), e.g. 6=2: four defs "
                    "with one group per workgroup; the line carries it in config.ctx_options")
    ap.add_argument("--planes-stand-in", action="store_true", help="--planes: the allocator chooses with launches over a constant stand-in input (hrx_alloc_output_planes) instead of this "
                    "run's batch (hrx_alloc_output_planes_for_batch)")
    ap.add_argument("--probes", action="store_true", help="also run the side probes earlier rounds' documents cite (all after the timed region): the same launches over ONE buffer set, "
                    "tools/mixceil, the input transposer alone and in front of the launch")
    ap.add_argument("--no-other-configs", action="store_true", help="default run only: skip the legs over BASELINE configs[2..4] (other_configs in the line)")
    ap.add_argument("--leg", action="store_true", help=argparse.SUPPRESS)     # one of the other_configs legs: a child of the default run
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)   # spawned by a bare --gpus N run
    args = ap.parse_args(argv)
    args.argv = [a for a in argv if a != "--child"]
    return args


# ----------------------------------------------------------------------------------------------------------------
# checker / baseline legs (the oracle is never on the product path)
# ----------------------------------------------------------------------------------------------------------------
def effective_cores():
    """Host cores this process can actually keep busy: its affinity mask, capped by the CPU bandwidth its cgroup grants (the GPU boxes of this pool show 256 cores under a quota of 16:
    256 threads run in a burst and are then throttled for the rest of the period)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except Exception:
        try:
            q, p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, -(-q // p)))
        except Exception:
            pass
    return n


def oracle_handle(names, allow_build=True):
    """the oracle (test infrastructure): None when its library is missing / stale and must not be built here (under a profiler
    no helper process may be started)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    if not allow_build:
        so = os.path.join(ROOT, "oracle", "_build", "libhrx_oracle.so")
        src = os.path.join(ROOT, "oracle", "hrx_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            return None
    return oracle_lib.OracleDefs(oracle_lib.load_oracle(), names)


def single_string(cfg, hra, o, chars, lens, M, calls=2000):
    """BASELINE configs[0]: the reference's own CPU-runnable case — ONE string per call, what `cargo test --release` exercises through
    RegexVerifyConfig::match_substrs (lib.rs:311-773, the test at lib.rs:1068).  The product serves it with the native host walk behind
    hrx_match_substrs (csrc/hrx_host_walk.cpp; no device work, no PCIe); the oracle's orc_match_substrs is timed beside it on the same
    string, both through ctypes with preallocated output columns (the foreign-call overhead, ~1-2 us, is in both figures)."""
    import ctypes as C
    import numpy as np
    b = int(np.argmax(lens))
    n = int(lens[b])
    ch = np.ascontiguousarray(chars[b, :n])
    D = cfg.num_defs
    u8p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)
    cols = [np.zeros(M, np.uint64), np.zeros(M, np.uint64)] + [np.zeros((D, M), np.uint64) for _ in range(4)] + [np.zeros(M, np.uint64), np.zeros(M, np.uint64)]
    ptrs = [c.ctypes.data_as(u64p) for c in cols]
    status = np.zeros(1, np.uint64)
    ctx, cp = cfg._need_ctx(), ch.ctypes.data_as(u8p)
    fn = hra.lib.hrx_match_substrs
    st = status.ctypes.data_as(u64p)
    for _ in range(20):
        rc = fn(ctx, cp, n, M, *ptrs, st)
    if rc != 0:
        return None
    t0 = time.perf_counter()
    for _ in range(calls):
        fn(ctx, cp, n, M, *ptrs, st)
    us = (time.perf_counter() - t0) / calls * 1e6
    out = {"us_per_call": us, "rows_per_s": n / (us * 1e-6), "n": n, "max_chars_size": M, "calls": calls, "cores": 1,
           "what": "hrx_match_substrs (the C-ABI stand-in for RegexVerifyConfig::match_substrs, lib.rs:311-773) on one %d-byte string of the batch, integer columns of "
                   "all %d witness rows out; the native host walk, one thread, timed through ctypes" % (n, M)}
    if o is not None:
        info = np.zeros(5, np.uint64)
        ofn, oh, ip = o.lib.orc_match_substrs, o.h, info.ctypes.data_as(u64p)
        for _ in range(5):
            ofn(oh, cp, n, M, *ptrs, ip)
        oc = max(50, calls // 10)
        t0 = time.perf_counter()
        for _ in range(oc):
            ofn(oh, cp, n, M, *ptrs, ip)
        ous = (time.perf_counter() - t0) / oc * 1e6
        out["oracle_us_per_call"] = ous
        out["oracle_what"] = "orc_match_substrs (oracle/hrx_oracle.c: the reference's hash-map / hash-set walk and the gate-by-gate mask recurrences on integers), same string, %d calls" % oc
    return out


def cpu_baseline(o, names, chars, lens, M, budget_s=8.0):
    """The oracle (oracle/hrx_oracle.c, the reference-faithful C port) timed single-threaded on a bounded sample of
    the same workload — the reference itself is single-threaded — plus, as extra context, the same port
    over all host cores and the oracle's dense-table "best CPU" variant.  Returns (json object, oracle outputs of the sample)."""
    import numpy as np
    nstr = min(len(chars), 16384)
    cores = effective_cores()
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


def batch_source_index(pos, k, shift, B, nd, sb):
    """Which of the nd distinct strings sits at position `pos` (a tensor of string positions) of buffer set k: set k is the batch rotated by
    k * shift strings (torch.roll: out[p] = in[(p - s) % B]) and the batch is ceil(B / nd) blocks of nd strings, block j = the distinct strings
    rotated by j * sb."""
    i = (pos - k * shift) % B
    return (i % nd - (i // nd) * sb) % nd


def verify_timed_buffers(o, hra, sets, shift, chars, lens, M, D, pm, dev, B, sb):
    """Bit-exact comparison of EVERY string of EVERY buffer set with the oracle's rows: the oracle walks the distinct strings of the batch once on
    all host cores (the whole batch unless --distinct), its rows are uploaded, and every string of set k — the batch rotated by k * shift strings — is
    compared on the device, where it lies, with the rows of the string it is a copy of (chunks of at most 2^28 record words at a time)."""
    import numpy as np
    import torch
    cores = effective_cores()
    nd = len(lens)
    orec, omsk, ost = o.witness_batch(chars, lens, M, threads=cores)
    ok = torch.from_numpy(((ost & np.uint64(0xff)) == 0)).to(dev)
    d_orec = torch.from_numpy(orec.view(np.int32)).to(dev)
    d_omsk = torch.from_numpy(omsk.view(np.int16)).to(dev)
    d_ost = torch.from_numpy(ost.view(np.int64)).to(dev)
    d_len = torch.from_numpy(lens.astype(np.int64)).to(dev)
    del orec, omsk
    q4, q8 = (M + 3) // 4, (M + 7) // 8
    step = max(1, min(hra.PM_BLOCK, (1 << 28) // max(1, M * D)))
    exact, rows, strings = True, 0, 0
    for k, (_, _, (rec, msk, st)) in enumerate(sets):
        for blk in range(0, B, hra.PM_BLOCK):                       # (position-major buffers are blocked by PM_BLOCK strings; string-major ones: just chunks)
            nb = min(hra.PM_BLOCK, B - blk)
            for a in range(0, nb, step):
                e = min(nb, a + step)
                if pm and isinstance(rec, (list, tuple)):      # record planes: def d's plane is [q4][nb][4] per block; one def in R row stripes: quad q in buffer q % R at slot q / R
                    R = len(rec) // D
                    slots = (q4 + R - 1) // R
                    per_def = []
                    for d_ in range(D):
                        qs = [rec[r_ * D + d_].view(-1)[blk * slots * 4:][:slots * nb * 4].view(slots, nb, 4)[:, a:e] for r_ in range(R)]
                        qq = qs[0] if R == 1 else torch.stack(qs, dim=1).reshape(slots * R, e - a, 4)[:q4]
                        per_def.append(qq.permute(1, 0, 2).reshape(e - a, q4 * 4)[:, :M])
                    r = torch.stack(per_def, dim=2)
                    m = msk.view(-1)[blk * q8 * 8:][:q8 * nb * 8].view(q8, nb, 8)[:, a:e].permute(1, 0, 2).reshape(e - a, q8 * 8)[:, :M]
                elif pm:
                    r = rec.view(-1)[blk * q4 * D * 4:][:q4 * D * nb * 4].view(q4, D, nb, 4)[:, :, a:e].permute(2, 0, 3, 1).reshape(e - a, q4 * 4, D)[:, :M]
                    m = msk.view(-1)[blk * q8 * 8:][:q8 * nb * 8].view(q8, nb, 8)[:, a:e].permute(1, 0, 2).reshape(e - a, q8 * 8)[:, :M]
                else:
                    r, m = rec[blk + a:blk + e], msk[blk + a:blk + e]
                idx = batch_source_index(torch.arange(blk + a, blk + e, device=dev), k, shift, B, nd, sb)
                okk = ok[idx]
                exact = exact and bool(torch.equal(st[blk + a:blk + e], d_ost[idx])) and bool(torch.equal(r[okk], d_orec[idx][okk])) \
                    and bool(torch.equal(m[okk], d_omsk[idx][okk]))
                rows += int(d_len[idx].sum())
                strings += e - a
                del idx, okk, r, m
    return {"strings": strings, "strings_per_set": int(B), "buffer_sets": len(sets), "rows": rows, "bit_exact": exact, "distinct_strings": int(nd),
            "against": "oracle/hrx_oracle.c (%d threads) on the %d distinct strings of the batch: status words, records and masked rows of EVERY string of every buffer set, "
                       "as written by one more replay of the timed graph over outputs poisoned after the timed region" % (cores, nd)}


def mix_ceiling(dev_index):
    """No-compute ceiling of the bench line's traffic mix on THIS box (tools/mixceil, built by __graft_entry__.build()):
    the kernel's own address streams (64 MiB read, 384 MiB written in position-major slabs) issued by 4 reader + 4 writer
    waves per CU with nothing else to do, and a plain dwordx4 copy of the same byte count.  Runs after the timed region."""
    exe = os.path.join(ROOT, "tools", "mixceil")
    if not os.path.exists(exe) or under_profiler():
        return None
    try:
        env = dict(os.environ, HIP_VISIBLE_DEVICES=physical_device(dev_index))
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


def profiled_now():
    return under_profiler()


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
    for flag in ("--steps", "--warmup", "--gpus", "--sets"):          # the child runs 3 eager launches on one device, one buffer set
        while flag in keep:
            i = keep.index(flag)
            del keep[i:i + 2]
    out = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            tmp = tempfile.mkdtemp(prefix="hrx_pmc_", dir="/tmp")
            env = dict(os.environ, TMPDIR="/tmp", HIP_VISIBLE_DEVICES=physical_device(dev_index))
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", tmp, "-o", "r1", "--", sys.executable, os.path.abspath(__file__)] + keep + \
                  ["--eager", "--steps", "3", "--warmup", "1", "--sets", "1", "--no-cpu-baseline", "--no-verify", "--no-spread", "--no-pmc"]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd="/tmp")
            vals = []
            for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "witness" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                        vals.append(float(row["Counter_Value"]))
            shutil.rmtree(tmp, ignore_errors=True)
            if r.returncode != 0 or not vals:
                sys.stderr.write("rocprofv3 --pmc %s pass failed (rc %d): roofline.traffic is null in this line\n" % (counter, r.returncode))
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
    elif args.config == "headers4":     # D = 4: the smallest config of the def-parallel launch over CLASS-WIDE tables
        hdr = lambda n, ns: (rd(n + "_lookup.txt"), [rd("%s_substr%d.txt" % (n, k)) for k in range(ns)])
        names, label = [hdr("header_from", 1), hdr("header_to", 1), hdr("header_subject", 3), pair(1)], "from/to/subject header definitions + regex1 (6 substrs)"
        gen = synth.headers_planted if args.dist == "planted" else synth.noise
    elif args.config == "headers5":     # D = 5: more than one launch walks side by side -> passes over groups of defs + combine
        hdr = lambda n, ns: (rd(n + "_lookup.txt"), [rd("%s_substr%d.txt" % (n, k)) for k in range(ns)])
        names, label = [hdr("header_from", 1), hdr("header_to", 1), hdr("header_subject", 3), pair(1), pair(2)], "from/to/subject header definitions + regex1 + regex2 (7 substrs, multi-pass)"
        gen = synth.headers_planted if args.dist == "planted" else synth.noise
    else:
        allb = np.arange(256, dtype=np.uint8)
        if args.substr_defs > 1:
            a_txt, subs = synth.random_dfa_multi(256, seed=2, alphabet=allb, n_substr_pairs=args.substr_pairs, n_substrs=args.substr_defs)
        else:
            a_txt, sub_txt = synth.random_dfa(256, seed=2, alphabet=allb, n_substr_pairs=args.substr_pairs)
            subs = [sub_txt]
        names, label, alphabet = [(a_txt.encode(), [t.encode() for t in subs])], "synthetic total DFA 256 states x 256 symbols (seed 2), %d substring definition%s of %d random tagged (state, next) pairs%s" % (
            len(subs), "" if len(subs) == 1 else "s", args.substr_pairs, "" if len(subs) == 1 else " each"), "all 256 byte values"
        gen = lambda B, n, seed=0, stride=None: synth.noise(B, n, seed=seed, alphabet=allb, stride=stride)
    planted = gen in (synth.regex1_planted, synth.regex23_planted, synth.headers_planted)
    return names, label, alphabet, gen, planted


def pin_to_device_numa_node(device_index):
    """Best effort, before any GPU call: run this rank on the host cores next to its GPU (the launch loop and the graph upload
    then do not cross sockets).  HIP enumerates devices in PCI bus order; the render nodes under /sys/class/drm carry the NUMA
    node of each AMD GPU.  Silently does nothing where the topology cannot be read."""
    try:
        import glob
        gpus = []
        for d in glob.glob("/sys/class/drm/renderD*/device"):
            if open(os.path.join(d, "vendor")).read().strip() == "0x1002":
                gpus.append((os.path.basename(os.path.realpath(d)), int(open(os.path.join(d, "numa_node")).read())))
        gpus.sort()
        vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
        idx = int(vis.split(",")[device_index]) if vis else device_index
        node = gpus[idx][1]
        if node < 0:
            return None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        if cpus:
            os.sched_setaffinity(0, cpus)
            return node
    except Exception:
        pass
    return None


def pick_sets(args, footprint_bytes):
    """How many input / output buffer sets the timed steps rotate over (--sets; default 8): with fewer than ~2 GB touched between two
    uses of a buffer the 256-MB Infinity Cache is part of the measurement.  Capped so that all sets fit 48 GiB."""
    n = max(1, args.sets)
    while n > 1 and n * footprint_bytes > (48 << 30):
        n -= 1
    return n


def run_rank(args, rank, world, device_index, barrier):
    """Everything one rank does; returns its result dict (rank 0's carries the line's descriptive fields)."""
    numa = pin_to_device_numa_node(device_index)
    import numpy as np
    import torch
    import halo2_regex_amd as hra

    dbg = os.environ.get("HRX_DEBUG_FLAGS", "")
    if dbg.strip() not in ("", "0", "0x0") and not args.allow_debug_flags:
        raise SystemExit("bench.py refuses to run with HRX_DEBUG_FLAGS=%r set: the timed kernel must be the one the planner picks" % dbg)
    default_lib = os.path.join(ROOT, "halo2_regex_amd", "csrc", "libhrx.so")
    if os.path.realpath(hra.LIB_PATH) != os.path.realpath(default_lib) and not args.allow_debug_flags:
        raise SystemExit("bench.py refuses to run with HRX_LIB_PATH=%r: the timed library must be the release build %s" % (hra.LIB_PATH, default_lib))
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    names, label, alphabet, gen, planted = workload(args)
    D = len(names)
    M, n = args.rows, args.n
    if args.scaling == "strong":      # the job is --batch strings in total, sharded by string index (hrx_shard_range)
        b_begin, B = hra.shard_range(args.batch, world, rank)
        if B == 0:
            raise SystemExit("rank %d: empty shard (--batch %d over %d ranks)" % (rank, args.batch, world))
    else:
        b_begin, B = rank * args.batch, args.batch
    pm = args.layout == "position-major"
    planes = bool(args.planes) and pm       # (one def: one records buffer from the planes allocator, or --stripes 2)
    rec_pitch, msk_pitch, rec_stride = hra.recommended_pitches(M)
    if args.dense or pm:
        rec_pitch, msk_pitch, rec_stride = M, M, (max(n, 1) + 15) // 16 * 16
    stride = rec_stride
    defs = [hra.RegexDefs(hra.AllstrRegexDef(a), [hra.SubstrRegexDef(t) for t in subs]) for a, subs in names]
    cfg = hra.RegexVerifyConfig.configure(M, defs, device=device_index)
    for kv in args.ctx_option:
        cfg.set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
    if under_profiler():
        cfg.set_option(hra.OPT_PLACE_DRY_LAUNCH, 0)      # the profiler's per-kernel averages then cover this script's launches only, not the allocator's launches into candidate sets

    # this rank's shard of the job: independent strings, seeded per rank (strong scaling: per shard start).  --distinct nd: nd generated strings,
    # the batch = ceil(B / nd) blocks of them, block j rotated by j * sb strings (built on the device)
    nd = min(B, args.distinct) if args.distinct > 0 else B
    sb = (nd // 7 + 11) % nd if nd < B else 0
    chars, lens = gen(nd, n, seed=rank if args.scaling == "weak" else 7919 * b_begin + world, stride=stride)
    d_lens0 = torch.from_numpy(lens.astype(np.int32)).to(dev)
    d_chars0 = torch.from_numpy(chars).to(dev)
    if nd < B:
        nblk = (B + nd - 1) // nd
        d_chars0 = torch.cat([torch.roll(d_chars0, shifts=j * sb, dims=0) if j else d_chars0 for j in range(nblk)])[:B].contiguous()
        d_lens0 = torch.cat([torch.roll(d_lens0, shifts=j * sb, dims=0) if j else d_lens0 for j in range(nblk)])[:B].contiguous()
    rows_per_step = int(d_lens0.sum(dtype=torch.int64))
    q4, q8 = (M + 3) // 4, (M + 7) // 8
    foot = B * stride + B * rec_pitch * 4 * D + B * msk_pitch * 2
    nsets = pick_sets(args, foot)
    # Buffer set k holds the same strings rotated by k * shift places (another arrangement of the same batch at other addresses:
    # generating 8 x 64 MiB of planted text on the host would take longer than everything else in this script); its oracle rows
    # are the rotated oracle rows of set 0.
    shift = (B // nsets + 37) % B if nsets > 1 else 0
    sets, placement, sm_sets = [], [], []
    keep_sm = pm and world == 1 and not args.no_spread and not args.leg and nsets * B * stride <= (4 << 30)
    for r in range(world):        # one rank at a time: the placement search times memory traffic
        if r == rank:
            for k in range(nsets):
                c_k = d_chars0 if k == 0 else torch.roll(d_chars0, shifts=k * shift, dims=0)
                l_k = d_lens0 if k == 0 else torch.roll(d_lens0, shifts=k * shift, dims=0)
                if pm:
                    if keep_sm:
                        sm_sets.append(c_k.contiguous())          # the reference's input shape (one contiguous string per row): for roofline.from_string_major_input
                    c_k = hra.chars_to_position_major(c_k)       # [stride/16][B][16]: done once, outside the timed region
                    # (--planes: the allocator's choosing launches run this very set's batch — hrx_alloc_output_planes_for_batch)
                    out = (cfg.alloc_output_planes(B, dev, stripes=args.stripes if D == 1 else None, chars=None if args.planes_stand_in else c_k, lens=l_k.contiguous(), chars_pm_stride=stride)
                           if planes else cfg.alloc_outputs_position_major(B, dev))
                else:
                    c_k = c_k.contiguous()
                    out = cfg.alloc_outputs(B, dev, pitched=not args.dense)
                rep = cfg.last_placement_report() if (sum(p_.numel() for p_ in out[0]) if planes else out[0].numel()) * 4 >= hra.PLACED_FROM else {"searched": 0}
                placement.append(rep)
                sets.append((c_k, l_k, out))
            torch.cuda.synchronize()
        if world > 1:
            barrier()
    del d_chars0
    if planes:
        launch = lambda i: cfg.witness_batch_planes(sets[i % nsets][0], sets[i % nsets][1], out=sets[i % nsets][2], chars_pm_stride=stride)
    elif pm:
        launch = lambda i: cfg.witness_batch_position_major(sets[i % nsets][0], sets[i % nsets][1], out=sets[i % nsets][2], chars_pm_stride=stride)
    else:
        launch = lambda i: cfg.witness_batch(sets[i % nsets][0], sets[i % nsets][1], out=sets[i % nsets][2])

    def sync_barrier():
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()

    # the kernel and geometry the planner picks for this shape on this device (what rocprofv3 will list)
    desc = cfg.describe_launch(B, layout=(3 | (hra.LAYOUT_RECORD_PLANES if planes else 0)) if pm else 0, num_cus=torch.cuda.get_device_properties(dev).multi_processor_count)

    def graph_of(fn, count):
        """`count` calls of fn(i) recorded once into a HIP graph (stream capture of the very same calls); None if capture fails"""
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                # thread_local: the RCCL watchdog thread of a multi-GPU run may poll events while this thread captures
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    for i in range(count):
                        fn(i)
            torch.cuda.current_stream(dev).wait_stream(side)
            return g
        except Exception as e:                              # capture unsupported: the caller falls back to plain launches
            sys.stderr.write("graph capture failed (%s): eager launches\n" % e)
            torch.cuda.synchronize()
            return None

    def timed_replays(run, reps, count):
        per = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record()
            torch.cuda.synchronize()
            per.append(e0.elapsed_time(e1) / count)
        return per

    t_warm = time.perf_counter()
    for i in range(args.warmup):
        launch(i)
    sync_barrier()
    warm_launches = args.warmup
    # The K timed steps are K kernel launches, step i over buffer set i % nsets.  They are recorded once into a HIP graph and the
    # graph is replayed inside the timed region, so that a slow host thread cannot turn the measurement into a launch-rate test
    # (one launch is ~80 us of device time); --eager launches them one by one.
    run_steps, launch_mode = None, "eager"
    if not args.eager:
        g = graph_of(launch, args.steps)
        if g is not None:
            # untimed replays of the same graph, ~100 ms of them (1300 launches of the bench line): graph upload, and the device's own ramp — with 100 launches (8 ms)
            # behind it a 20-step replay averaged 80.0 us per launch, with 1000 launches 77.9, and from K = 100 on the count no longer matters
            # (profiles/r03_probes/warmup_ramp.txt; a kernel trace shows the launches of a replay that follows a pause taking 92 -> 84 us over
            # its first twenty, profiles/r03_pm_trace_replays.txt); W = 5 eager steps do not cover that.  The line says so: `warmup_effective`.
            g.replay(); torch.cuda.synchronize()            # (upload)
            tr = time.perf_counter(); g.replay(); torch.cuda.synchronize(); tr = max(time.perf_counter() - tr, 1e-5)
            untimed = args.untimed_replays if args.untimed_replays > 0 else max(2, min(1000, int(0.1 / tr) + 1))    # ~100 ms of load
            for _ in range(untimed - 2):
                g.replay()
            torch.cuda.synchronize()
            warm_launches += untimed * args.steps
            run_steps, launch_mode = g.replay, "hipGraph of %d kernel nodes (after %d untimed replays of it = ~100 ms; the end of the timed replay is awaited by spinning on its end event)" % (args.steps, untimed)
    if run_steps is None:
        def run_steps():
            for i in range(args.steps):
                launch(i)   # launched on torch's current stream, where the events sit
    # (Round 3 poisoned the outputs HERE, between the untimed replays and the timed one: the timed replay then was the first thing after
    # a burst of strided fills and read 5 % slower than the replays that follow it.  The poison now comes AFTER the timed region, in front
    # of one more replay of the very same graph, and that replay's outputs are what the verification reads — see below.)
    sync_barrier()
    warm_ms = (time.perf_counter() - t_warm) * 1e3
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run_steps()                                            # EXACTLY K steps
    ev1.record()
    while not ev1.query():                                 # (spin: a blocking wait's wake-up occasionally costs 100+ us — 10 % of a 20-step region)
        pass
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    kern_ms = ev0.elapsed_time(ev1) / args.steps      # average launch duration, HIP events on the launch stream

    # What the verification reads must have been written by the timed work: poison the outputs — status words entirely; records and
    # masked rows at one element per ~64 KiB (filling the 3 GB of buffers outright would take seconds of host-visible time for nothing:
    # a launch that skipped a string, a tile or a plane shows at these spots as surely) — and replay the SAME executable graph (or the same
    # eager sequence) once more.  Every launch of the timed region is one of this replay's launches, with the same arguments.
    for _, _, out in sets:
        if pm:                                 # (prime strides: every string and every row position gets its share of the spots)
            for r_ in (out[0] if planes else [out[0]]):
                r_.view(-1)[::16411].fill_(-1)
            out[1].view(-1)[::32771].fill_(-1)
        else:                                  # (string-major outputs are pitched views: every 64th row of every string)
            out[0][:, ::64].fill_(-1)
            out[1][:, ::64].fill_(-1)
        out[2].fill_(-1)
    torch.cuda.synchronize()
    run_steps()
    torch.cuda.synchronize()
    # sanity: every string of every buffer set the timed steps wrote finished with status 0
    written = min(nsets, args.steps)
    for k in range(written):
        st = sets[k][2][2].cpu().numpy().view(np.uint64)
        assert ((st & np.uint64(0xff)) == 0).all(), "status != ok in the bench workload (buffer set %d)" % k

    res = {"rank": rank, "device": device_index, "rows": rows_per_step * args.steps, "elapsed_s": elapsed, "avg_launch_ms": kern_ms,
           "debug_flags": dbg or None, "numa_node": numa, "strings": B, "shard_begin": b_begin,
           "warmup_effective": {"launches": warm_launches, "eager_steps": args.warmup, "graph_replays": (warm_launches - args.warmup) // max(args.steps, 1),
                                "ms": warm_ms, "note": "every launch of this workload that ran before the timed region (the --warmup eager steps + the untimed replays of the K-step graph), "
                                                       "and the wall time from the first of them to the barrier in front of the timed region (includes graph capture)"}}
    profiled = under_profiler()
    o = None
    if (not args.no_verify and (rank == 0 or args.verify_all_ranks)) or (rank == 0 and not args.no_cpu_baseline):
        o = oracle_handle(names, allow_build=not profiled)
    if not args.no_verify and (rank == 0 or args.verify_all_ranks):
        if o is None:
            res["verified"] = {"bit_exact": None, "skipped": "under a profiler and the oracle library is not built: no helper process may be started here"}
        else:
            res["verified"] = verify_timed_buffers(o, hra, sets[:written], shift, chars, lens, M, D, pm, dev, B, sb)
            if not res["verified"]["bit_exact"]:
                raise SystemExit("bench.py: the timed output buffers differ from the oracle")
    if rank != 0:
        return res

    # spread: R more replays of the same K steps, each timed by its own event pair (after the contract's timed region)
    if not args.no_spread:
        reps = max(5, min(100, int(0.25 / max(kern_ms * args.steps * 1e-3, 1e-6))))
        per = timed_replays(run_steps, reps, args.steps)
        res["spread"] = {"replays": reps, "steps_per_replay": args.steps, "ms_per_step_median": statistics.median(per),
                         "ms_per_step_min": min(per), "ms_per_step_max": max(per)}
    # per buffer set: three back-to-back launches over set k alone (multi-GB sets: where a set's buffers lie shows here; sets that fit the Infinity Cache: not an HBM figure)
    if not args.no_spread and world == 1 and nsets > 1 and foot >= (1 << 30):
        try:
            pss = []
            for k in range(nsets):
                launch(k); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    launch(k)
                e1.record(); torch.cuda.synchronize()
                pss.append(e0.elapsed_time(e1) / 3)
            res["per_set_ms"] = pss
        except Exception as e:
            sys.stderr.write("per-set timing failed: %s\n" % e)
    # The complementary figure: the same K launches re-processing ONE batch into ONE set of buffers (round 1 and 2's step).  From the
    # second launch on the 256-MB Infinity Cache holds part of what a launch reads and overwrites — not an HBM figure.
    if args.probes and not args.no_spread and world == 1 and nsets > 1:
        try:
            one = lambda i: launch(0)
            g1 = graph_of(one, args.steps) if not args.eager else None
            run1 = g1.replay if g1 is not None else (lambda: [one(i) for i in range(args.steps)])
            run1(); torch.cuda.synchronize()
            per = timed_replays(run1, 5, args.steps)
            res["one_buffer_set"] = {"ms_per_step_median": statistics.median(per), "ms_per_step_min": min(per), "ms_per_step_max": max(per)}
            del g1
        except Exception as e:                                   # a probe must never break the bench line
            sys.stderr.write("one-buffer-set probe failed: %s\n" % e)
    # No-compute ceiling on THESE buffers (after the verification: it overwrites the outputs): hrx_traffic_pass_device moves the bytes
    # of one launch — same addresses, same instructions, same store policy — and does no DFA work.
    if not args.no_spread and world == 1 and (pm or M % 8 == 0):
        try:
            tp = (lambda i: cfg.traffic_pass_planes(sets[i % nsets][0], B, sets[i % nsets][2], stride)) if planes else \
                 (lambda i: cfg.traffic_pass(sets[i % nsets][0], B, sets[i % nsets][2], stride)) if pm else \
                 (lambda i: cfg.traffic_pass_string_major(sets[i % nsets][0], sets[i % nsets][2]))
            gt = graph_of(tp, args.steps) if not args.eager else None
            runt = gt.replay if gt is not None else (lambda: [tp(i) for i in range(args.steps)])
            runt(); torch.cuda.synchronize()
            per = timed_replays(runt, 5, args.steps)
            mc = {"rotating_us": statistics.median(per) * 1e3}
            if nsets > 1 and args.probes:
                tp1 = (lambda i: cfg.traffic_pass_planes(sets[0][0], B, sets[0][2], stride)) if planes else \
                      (lambda i: cfg.traffic_pass(sets[0][0], B, sets[0][2], stride)) if pm else (lambda i: cfg.traffic_pass_string_major(sets[0][0], sets[0][2]))
                gt1 = graph_of(tp1, args.steps) if not args.eager else None
                runt1 = gt1.replay if gt1 is not None else (lambda: [tp1(i) for i in range(args.steps)])
                runt1(); torch.cuda.synchronize()
                mc["one_set_us"] = statistics.median(timed_replays(runt1, 5, args.steps)) * 1e3
                del gt1
            res["traffic_pass"] = mc
            del gt
        except Exception as e:
            sys.stderr.write("traffic-pass probe failed: %s\n" % e)
    # Starting from the reference's input shape — B contiguous strings (lib.rs:311-315) instead of the position-major chunks the headline reads:
    # (a) hrx_chars_to_position_major_device alone, (b) that + the same launch (what a caller holding &[u8] strings pays for the fast path),
    # (c) the position-major kernel reading the strings directly (HRX_LAYOUT_POSITION_MAJOR without HRX_LAYOUT_INPUT_POSITION_MAJOR: 16 bytes per lane, a stride apart).
    if sm_sets and world == 1:
        try:
            fsm = {}
            tr = lambda i: cfg.chars_to_position_major_device(sm_sets[i % nsets], out=sets[i % nsets][0])
            both = lambda i: (tr(i), launch(i))
            direct = lambda i: cfg.witness_batch_position_major(sm_sets[i % nsets], sets[i % nsets][1], out=sets[i % nsets][2])
            for key, fn in ((("transpose", tr), ("transpose_plus_launch", both)) if args.probes else ()) + (("string_major_input_launch", direct),):
                gg = graph_of(fn, args.steps) if not args.eager else None
                rr = gg.replay if gg is not None else (lambda fn=fn: [fn(i) for i in range(args.steps)])
                rr(); torch.cuda.synchronize()
                fsm[key + "_ms"] = statistics.median(timed_replays(rr, 5, args.steps))
                if key == "string_major_input_launch" and o is not None and not planes:
                    # the rows this route writes, compared with the oracle like the headline's: outputs poisoned, the same graph replayed once more, every string of every set checked
                    for _, _, out_ in sets:
                        out_[0].view(-1)[::16411].fill_(-1); out_[1].view(-1)[::32771].fill_(-1); out_[2].fill_(-1)
                    torch.cuda.synchronize()
                    rr(); torch.cuda.synchronize()
                    v_ = verify_timed_buffers(o, hra, sets[:min(nsets, args.steps)], shift, chars, lens, M, D, pm, dev, B, sb)
                    fsm["string_major_input_launch_verified"] = {"bit_exact": v_["bit_exact"], "strings": v_["strings"], "buffer_sets": v_["buffer_sets"]}
                del gg
            # the transposer leaves the very bytes the headline launch read: the outputs after (b) are the verified ones
            st_ok = all(((sets[k][2][2].cpu().numpy().view(np.uint64) & np.uint64(0xff)) == 0).all() for k in range(min(nsets, args.steps)))
            fsm["status_ok_after"] = bool(st_ok)
            fsm["kernel_of_string_major_input_launch"] = cfg.describe_launch(B, layout=1, num_cus=torch.cuda.get_device_properties(dev).multi_processor_count).split(" grid=")[0]
            res["from_string_major_input"] = fsm
        except Exception as e:
            sys.stderr.write("string-major-input probe failed: %s\n" % e)
    del sm_sets
    # The path an UNMODIFIED caller of the seam takes: host Vecs in, host Vecs out (lib.rs:311-318) through hrx_witness_batch_host — PCIe-inclusive, never `value` —
    # and what a host consumer pays to pull ONE circuit's rows out of position-major buffers copied to the host as they are.
    if is_default_workload(args) and world == 1 and not args.no_spread and not profiled_now():
        try:
            import ctypes as C
            hrec, hmsk, hst = np.empty((B, M, D), np.uint32), np.empty((B, M), np.uint16), np.empty(B, np.uint64)
            hc = np.ascontiguousarray(chars[:, :stride])
            cfg.witness_batch_host(hc, lens, out=(hrec, hmsk, hst))          # (first call: the context's staging buffers, first touch of the output pages)
            # The three routes of hrx_witness_batch_host (HRX_OPT_HOST_ROUTE), same batch, same pageable arrays: everything through the device (staged, walked, copied back: the link
            # bounds it), everything on the host cores (the native walk on every core this process may run on), and the default — both at once, split by the measured rates.
            def timed_calls(n_warm, n):
                for _ in range(n_warm):
                    cfg.witness_batch_host(hc, lens, out=(hrec, hmsk, hst))
                ts_ = []
                for _ in range(n):
                    t0_ = time.perf_counter()
                    cfg.witness_batch_host(hc, lens, out=(hrec, hmsk, hst))
                    ts_.append(time.perf_counter() - t0_)
                return ts_
            cfg.set_option(hra.OPT_HOST_ROUTE, hra.HOST_ROUTE_DEVICE)
            explore = [t * 1e3 for t in timed_calls(0, 4)]      # the context's own comparison of its two transfer modes: pipelined, one stream, pipelined, one stream (hrx_host_api.cpp) — timed, reported apart
            t_dev = timed_calls(0, 5)
            dev_rep = cfg.host_route_report()
            cfg.set_option(hra.OPT_HOST_ROUTE, hra.HOST_ROUTE_HOST)
            t_host = timed_calls(1, 3)
            host_rep = cfg.host_route_report()
            cfg.set_option(hra.OPT_HOST_ROUTE, hra.HOST_ROUTE_AUTO)
            cfg_auto = cfg.clone()                              # (a context of its own: its calls 0-4 are the measuring ones — device, device, host cores, split, split)
            cfg_auto.set_option(hra.OPT_HOST_ROUTE, hra.HOST_ROUTE_AUTO)
            t_explore = []
            for _ in range(6):
                t0_ = time.perf_counter(); cfg_auto.witness_batch_host(hc, lens, out=(hrec, hmsk, hst)); t_explore.append((time.perf_counter() - t0_) * 1e3)
            ts = []
            for _ in range(5):
                t0_ = time.perf_counter(); cfg_auto.witness_batch_host(hc, lens, out=(hrec, hmsk, hst)); ts.append(time.perf_counter() - t0_)
            auto_rep = cfg_auto.host_route_report()
            ms = statistics.median(ts) * 1e3
            best_single = min(statistics.median(t_dev), statistics.median(t_host)) * 1e3
            e2e = {"ms_per_call": ms, "ms_min": min(ts) * 1e3, "rows_per_s": rows_per_step / (ms * 1e-3), "calls": 5,
                   "bytes_out": int(hrec.nbytes + hmsk.nbytes + hst.nbytes), "bytes_in": int(hc.nbytes + 4 * B),
                   "gbs_out": (hrec.nbytes + hmsk.nbytes + hst.nbytes) / (ms * 1e-3) / 1e9,
                   "status_ok": bool(((hst & np.uint64(0xff)) == 0).all()),
                   "routes": {"auto_ms": ms, "device_ms": statistics.median(t_dev) * 1e3, "host_walk_all_cores_ms": statistics.median(t_host) * 1e3,
                              "auto_over_best_single_route": ms / best_single,
                              "auto_chose": {0: "split", 1: "device", 2: "host cores"}.get(auto_rep["route"]), "auto_measuring_calls_ms": t_explore,
                              "auto_report": {k: auto_rep[k] for k in ("route", "device_strings", "host_strings", "device_ms", "host_ms", "device_alone_ns_per_row", "host_alone_ns_per_row",
                                                                       "split_ns_per_row", "device_ns_per_row", "host_ns_per_row", "host_threads", "device_pipelined")},
                              "host_threads": host_rep["host_threads"], "device_pipelined": dev_rep["device_pipelined"]},
                   "comparison_calls_ms": {"pipelined": [explore[0], explore[2]], "one_stream": [explore[1], explore[3]]},
                   "what": "hrx_witness_batch_host on the same batch: pageable host arrays in (string-major, %d B apart) and out (records [B][M][D] u32, masked [B][M] u16, status), "
                           "output arrays reused.  ms_per_call = the DEFAULT route (HRX_HOST_ROUTE_AUTO: the fastest of device / host cores / both at once by the context's own measurements — "
                           "its calls 0-4 measure: auto_measuring_calls_ms; auto_chose says which way the timed calls went); routes.device_ms = everything staged, walked and copied back (chunk by chunk on two streams, or in, walk, out on one: "
                           "comparison_calls_ms are the context's own calls 2-5 that pick; the call lasts about as long as the copy out over the PCIe link); "
                           "routes.host_walk_all_cores_ms = the native walk on every core this process may run on (a rank is pinned to its GPU's NUMA node)" % stride}
            # what the link gives a plain device-to-host copy of the same bytes into the same (pageable, already touched) arrays on THIS box
            try:
                tr_, tm_ = torch.from_numpy(hrec.view(np.int32).reshape(-1)), torch.from_numpy(hmsk.view(np.int16).reshape(-1))
                dr_ = torch.empty(tr_.numel(), dtype=torch.int32, device=dev) if planes else sets[0][2][0].view(-1)[:tr_.numel()]
                dm_ = sets[0][2][1].view(-1)[:tm_.numel()]
                tc = []
                for _ in range(3):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    tr_.copy_(dr_); tm_.copy_(dm_)
                    torch.cuda.synchronize(); tc.append(time.perf_counter() - t0)
                e2e["copy_out_alone_ms"] = min(tc) * 1e3
                e2e["copy_out_alone_gbs"] = (tr_.numel() * 4 + tm_.numel() * 2) / min(tc) / 1e9
                e2e["copy_out_alone_what"] = "a plain device-to-host copy of as many bytes (records + masked rows) into the same host arrays, nothing else: the link's rate for this process on this box"
                cfg.witness_batch_host(hc, lens, out=(hrec, hmsk, hst))      # (the arrays hold the host path's rows again: compared with the gather below)
            except Exception as e:
                sys.stderr.write("copy-out probe failed: %s\n" % e)
            mp = sets[0][2][1].cpu().numpy().view(np.uint16)
            r1, m1 = np.empty((M, D), np.uint32), np.empty(M, np.uint16)
            picks = np.random.default_rng(0).integers(0, B, 4000)
            if planes:
                hps = [p_.cpu().numpy().view(np.uint32) for p_ in sets[0][2][0]]
                arr_ = (C.c_void_p * len(hps))(*[p_.ctypes.data for p_ in hps])
                fn = hra.lib.hrx_rows_of_string_planes
                args_c = (arr_, len(hps), mp.ctypes.data, B, M, D)
            else:
                rp = sets[0][2][0].cpu().numpy().view(np.uint32)
                fn = hra.lib.hrx_rows_of_string_position_major
                args_c = (rp.ctypes.data, mp.ctypes.data, B, M, D)
            same = True
            for b in picks[:64]:
                fn(*args_c, int(b), r1.ctypes.data, m1.ctypes.data)
                same = same and np.array_equal(r1, hrec[int(b)]) and np.array_equal(m1, hmsk[int(b)])    # (set 0 holds the batch unrotated)
            t0 = time.perf_counter()
            for b in picks:
                fn(*args_c, int(b), r1.ctypes.data, m1.ctypes.data)
            e2e["gather_us_per_string"] = (time.perf_counter() - t0) / len(picks) * 1e6
            e2e["gather_equals_host_path_rows"] = bool(same)
            e2e["gather_what"] = ("hrx_rows_of_string_position_major (record planes / row stripes: hrx_rows_of_string_planes): one circuit's [M][D] records + [M] masked rows gathered on one host core out of position-major HOST "
                                  "buffers (the device buffers copied out as they are), random strings, through ctypes")
            res["end_to_end_host"] = e2e
            del hrec, hmsk, hst, mp
        except Exception as e:
            sys.stderr.write("end-to-end host probe failed: %s\n" % e)
    # What the placement-aware allocation buys on THIS box: the same K steps, same inputs, into output buffers from two plain allocations per set (HRX_PLACE_OFF).
    if is_default_workload(args) and world == 1 and not args.no_spread and pm and not planes:
        try:
            cfg.set_placement(walk=False)
            plain_sets = [cfg.alloc_outputs_position_major(B, dev) for _ in range(nsets)]
            cfg.set_placement(walk=True)
            pl_fn = lambda i: cfg.witness_batch_position_major(sets[i % nsets][0], sets[i % nsets][1], out=plain_sets[i % nsets], chars_pm_stride=stride)
            gp = graph_of(pl_fn, args.steps) if not args.eager else None
            runp = gp.replay if gp is not None else (lambda: [pl_fn(i) for i in range(args.steps)])
            runp(); torch.cuda.synchronize()
            for _ in range(max(2, int(0.05 / max(kern_ms * args.steps * 1e-3, 1e-6)))):
                runp()
            torch.cuda.synchronize()
            per = timed_replays(runp, 5, args.steps)
            st_ok = all(((plain_sets[k][2].cpu().numpy().view(np.uint64) & np.uint64(0xff)) == 0).all() for k in range(min(nsets, args.steps)))
            res["plain_allocations"] = {"ms_per_step_median": statistics.median(per), "ms_per_step_min": min(per), "status_ok": bool(st_ok)}
            del gp, plain_sets
        except Exception as e:
            sys.stderr.write("plain-allocations probe failed: %s\n" % e)
    # SURVEY §8 f4 on the headline batch: the compact rows of 8192 strings per call -> bn256::Fr cells (hrx_fr_columns_device), timed; the cells of 64 strings checked against
    # hrx_fr_from_u64 of the integers hrx_witness_columns_host reads out of the same buffers copied to the host.
    if is_default_workload(args) and world == 1 and not args.no_spread and pm and not planes:
        try:
            NB = min(8192, B)
            launch(0); torch.cuda.synchronize()
            c0, l0, o0 = sets[0]
            for _ in range(2):
                cells = cfg.fr_columns(c0, l0, o0, b_begin=0, b_count=NB, position_major=True, chars_pm_stride=stride)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            e0.record()
            for i in range(reps):
                cells = cfg.fr_columns(c0, l0, o0, b_begin=(i * NB) % max(B - NB + 1, 1), b_count=NB, position_major=True, chars_pm_stride=stride)
            e1.record(); torch.cuda.synchronize()
            ms_fr = e0.elapsed_time(e1) / reps
            ncol = 4 + 4 * D
            nchk = 64
            mont = cfg.fr_columns(c0, l0, o0, b_begin=0, b_count=nchk, position_major=True, chars_pm_stride=stride).cpu().numpy().view(np.uint64)
            canon = cfg.fr_columns(c0, l0, o0, b_begin=0, b_count=nchk, position_major=True, chars_pm_stride=stride, canonical=True).cpu().numpy().view(np.uint64)
            cols = hra.witness_columns_host(c0.cpu().numpy(), l0.cpu().numpy().view(np.uint32), o0[0].cpu().numpy(), o0[1].cpu().numpy(), M, D, b_begin=0, b_count=nchk,
                                            position_major=True, chars_pm_stride=stride, B=B)
            ok_int = bool(np.array_equal(canon[..., 0], cols) and not canon[..., 1:].any())
            vals = np.unique(cols)
            table = {int(v): hra.fr_from_u64(int(v)) for v in vals}
            want = np.zeros(mont.shape, np.uint64)
            for v, limbs in table.items():
                want[cols == v] = np.asarray(limbs, np.uint64)
            ok_mont = bool(np.array_equal(mont, want))
            res["fr_columns"] = {"ms_per_call": ms_fr, "strings_per_call": NB, "cells_per_s": NB * M * ncol / (ms_fr * 1e-3), "written_GBps": NB * M * ncol * 32 / (ms_fr * 1e-3) / 1e9,
                                 "frac_of_peak": NB * M * (ncol * 32 + BYTES_PER_ROW(D)) / (ms_fr * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "checked_cells": int(mont.size // 4), "distinct_values": int(len(vals)),
                                 "canonical_cells_equal_hrx_witness_columns_host": ok_int, "montgomery_cells_equal_hrx_fr_from_u64": ok_mont,
                                 "what": "hrx_fr_columns_device (SURVEY §8 f4) on the headline batch: %d strings x %d rows x %d columns of 32-byte bn256::Fr cells per call out of the "
                                         "position-major witness buffers; the cells of strings 0..%d compared with hrx_fr_from_u64 of the integers hrx_witness_columns_host gives for the same "
                                         "buffers copied to the host (parity of the field arithmetic itself: big-integer arithmetic + halo2curves' published constants, tests/test_fr.py)" % (NB, M, ncol, nchk - 1)}
            del cells, mont, canon
            torch.cuda.empty_cache()
        except Exception as e:
            sys.stderr.write("fr_columns probe failed: %s\n" % e)
    # Record planes: the same K steps over the INTERLEAVED layout ([M/4][D][B][4] in one allocation + placed masked rows: what every earlier round measured), as many buffer
    # sets, in this process on this box — what the planes buy.  The planes' sets are released first (the verification and the no-compute pass are done with them).
    if planes and world == 1 and not args.no_spread:
        try:
            ins = [(c, l) for c, l, _ in sets]
            for k in range(len(sets)):
                sets[k] = (sets[k][0], sets[k][1], None)
            torch.cuda.empty_cache()
            il_sets = [cfg.alloc_outputs_position_major(B, dev) for _ in range(nsets)]
            il = lambda i: cfg.witness_batch_position_major(ins[i % nsets][0], ins[i % nsets][1], out=il_sets[i % nsets], chars_pm_stride=stride)
            gi = graph_of(il, args.steps) if not args.eager else None
            runi = gi.replay if gi is not None else (lambda: [il(i) for i in range(args.steps)])
            runi(); torch.cuda.synchronize()
            per = timed_replays(runi, 5, args.steps)
            st_ok = all(((il_sets[k][2].cpu().numpy().view(np.uint64) & np.uint64(0xff)) == 0).all() for k in range(min(nsets, args.steps)))
            res["interleaved_layout"] = {"ms_per_step_median": statistics.median(per), "ms_per_step_min": min(per), "status_ok": bool(st_ok),
                                         "kernel": cfg.describe_launch(B, layout=3, num_cus=torch.cuda.get_device_properties(dev).multi_processor_count).split(" grid=")[0]}
            del gi, il_sets
        except Exception as e:                                   # a probe must never break the bench line
            sys.stderr.write("interleaved-layout comparison failed: %s\n" % e)
    res["desc"] = desc
    res["placement"] = placement
    res["library"] = os.path.realpath(hra.LIB_PATH)
    res["nsets"] = nsets
    res["config"] = {"workload": "%s DFA (D=%d), %d x %d-byte strings per GPU (n=%d chars, M=%d witness rows), %s"
                                 % (label, D, B, stride, n, M, "uniform noise over the %s%s%s" % (alphabet, " + planted match" if planted else "",
                                    "" if nd == B else "; %d distinct strings, the batch = %d blocks of them, block j rotated by %d j strings" % (nd, (B + nd - 1) // nd, sb))),
                     "batch_per_gpu": B, "n": n, "max_chars_size": M, "defs": D, "rows_counted": "sum of n (character positions)", "ctx_options": list(args.ctx_option),
                     "buffers": ("HRX_LAYOUT_POSITION_MAJOR | HRX_LAYOUT_INPUT_POSITION_MAJOR with RECORD PLANES (blocks of 65536 strings): chars [%d/16][B][16], %s, "
                                 "masked [M/8][B][8] (include/hrx.h hrx_witness_batch_device_planes); the %d + 1 output buffers from "
                                 "hrx_alloc_output_planes (each in a neighbourhood of the device memory of its own)" % (stride, "every def's records [M/4][B][4] in a buffer of its own" if D > 1 else
                                 ("the one def's records in two ROW STRIPES [M/8][B][4] (quad q of a string in stripe q % 2 at slot q / 2)" if args.stripes == 2 else "the one def's records [M/4][B][4]"),
                                 D if D > 1 else args.stripes)) if planes else
                                ("HRX_LAYOUT_POSITION_MAJOR | HRX_LAYOUT_INPUT_POSITION_MAJOR (blocks of 65536 strings): chars [%d/16][B][16], records "
                                 "[M/4][D][B][4], masked [M/8][B][8] (include/hrx.h); outputs from hrx_alloc_outputs_position_major (placement-aware "
                                 "from 128 MiB of records on, two plain allocations below)" % stride) if pm else
                                ("string-major; input stride %d B, records pitch %d rows, masked pitch %d rows" % (stride, rec_pitch, msk_pitch)),
                     "buffer_sets": "%d input / output buffer sets, step i runs on set i %% %d (set k = the batch rotated by %d k strings): nothing a launch touches "
                                    "was touched by the previous %d launches, so the 256-MB Infinity Cache holds none of it" % (nsets, nsets, shift, nsets - 1)
                                    if nsets > 1 else "one buffer set, re-processed every step",
                     "sharding": "by string index, no collective", "launch_mode": launch_mode}
    res["D"], res["rows_per_step"] = D, rows_per_step
    if world == 1:
        del sets
        torch.cuda.empty_cache()
        if args.probes and args.config == "regex1" and B == 65536 and M == 1024 and pm:
            res["mix_ceiling"] = mix_ceiling(device_index)
        if not args.no_pmc:
            res["traffic"] = measured_traffic(args.argv, device_index)
    if not args.no_cpu_baseline and o is not None:
        res["cpu_baseline"] = cpu_baseline(o, names, chars, lens, M)
    if not args.no_cpu_baseline and world == 1:
        try:
            res["single_string"] = single_string(cfg, hra, o, chars, lens, M)
        except Exception as e:                                   # a side figure must never break the bench line
            sys.stderr.write("single-string timing failed: %s\n" % e)
    return res


# ----------------------------------------------------------------------------------------------------------------
# aggregation (imported by tests/test_dist_cpu.py)
# ----------------------------------------------------------------------------------------------------------------
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
    nsets = r0.get("nsets", 1)
    line = {
        "metric": "DFA witness rows/sec", "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": getattr(args, "scaling", "weak"), "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": r0["config"],
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": (r0.get("traffic") or {}).get("total"),   # null when the two rocprofv3 --pmc child passes did not run (--no-pmc, N > 1) or failed: never a number from another run
                     "kernel": r0["desc"].split(" grid=")[0], "launch": "grid=" + r0["desc"].split(" grid=")[1],
                     "avg_launch_ms": kern_ms,
                     "algorithmic_bytes_per_launch": algo_bytes, "bytes_per_row": BYTES_PER_ROW(D),
                     "regime": ("the K timed steps rotate over %d input / output buffer sets: every byte a launch moves comes from / goes to HBM" % nsets) if nsets > 1
                               else "the K timed steps re-process one batch into one set of buffers (part of the traffic stays in the 256-MB Infinity Cache)"},
        "ranks_seen": world,
        "per_rank": [dict({"rank": r["rank"], "device": r["device"], "rows": r["rows"], "elapsed_s": r["elapsed_s"],
                           "rows_per_s": r["rows"] / r["elapsed_s"], "avg_launch_ms": r["avg_launch_ms"]},
                          **({"strings": r["strings"], "shard_begin": r["shard_begin"]} if "strings" in r else {}),
                          **({"verified": r["verified"]["bit_exact"]} if r.get("verified") and r["rank"] != 0 else {})) for r in per_rank],
        "debug_flags": r0.get("debug_flags"),
    }
    if r0.get("library"):
        line["library"] = r0["library"]
    line["roofline"]["traffic_source"] = r0["traffic"] if r0.get("traffic") else None
    if r0.get("verified"):
        line["verified"] = r0["verified"]
    if r0.get("spread"):
        line["spread"] = r0["spread"]
    if r0.get("placement"):
        pl = r0["placement"]
        srch = [p for p in pl if p.get("searched")]
        line["roofline"]["placement"] = {
            "sets_searched": len(srch), "sets_accepted": sum(1 for p in srch if p.get("accepted")),
            "steps": [p.get("steps") for p in srch], "chosen_step": [p.get("chosen_step") for p in srch],
            "ref_gbs": [round(p.get("ref_gbs", 0)) for p in srch], "first_gbs": [round(p.get("first_gbs", 0)) for p in srch],
            "best_gbs": [round(p.get("best_gbs", 0)) for p in srch], "search_ms": [round(p.get("search_ms", 0), 1) for p in srch],
            "capped": [p.get("capped", 0) for p in srch],
            "what": "hrx_alloc_outputs_position_major per buffer set: two-stream probe rates (GB/s written, device clock) of the same-block reference, the first candidate "
                    "(= two plain allocations) and the kept masked-row buffer; accepted = kept one >= 10 % above the reference (DESIGN.md §6).  RECORD PLANES "
                    "(hrx_alloc_output_planes): steps = pairings measured, ref_gbs / best_gbs = the slowest pairing seen / of the kept set, first_gbs = of the first D + 1 "
                    "buffers (plain allocations), chosen_step = output bytes per row (of 4 D + 2) in the kept set's busiest class of the address space"}
    if r0.get("per_set_ms"):
        line["roofline"]["per_set_ms"] = [round(x, 4) for x in r0["per_set_ms"]]
    if r0.get("plain_allocations"):
        pa = r0["plain_allocations"]
        gbs = algo_bytes / (pa["ms_per_step_median"] * 1e-3) / 1e9
        line["roofline"]["plain_allocations"] = dict(pa, achieved=gbs, frac=gbs / HBM_PEAK_GBS,
                                                     what="the same K steps in this process into output buffers from two plain allocations per set (hrx_ctx_set_placement HRX_PLACE_OFF) "
                                                          "instead of the placement-aware ones: what the placement buys on this box")
    if r0.get("fr_columns"):
        line["fr_columns"] = r0["fr_columns"]
    if r0.get("interleaved_layout"):
        il = r0["interleaved_layout"]
        gbs = algo_bytes / (il["ms_per_step_median"] * 1e-3) / 1e9
        line["roofline"]["interleaved_layout"] = dict(il, achieved=gbs, frac=gbs / HBM_PEAK_GBS,
                                                      what="the same K steps in this process with the records INTERLEAVED in one allocation ([M/4][D][B][4] from "
                                                           "hrx_alloc_outputs_position_major: every earlier round's layout) over as many buffer sets — what the record planes buy on this box")
    if r0.get("one_buffer_set"):
        ob = r0["one_buffer_set"]
        gbs = algo_bytes / (ob["ms_per_step_median"] * 1e-3) / 1e9
        line["roofline"]["one_buffer_set"] = dict(ob, achieved=gbs, frac=gbs / HBM_PEAK_GBS,
                                                  what="the same K launches re-processing ONE batch into ONE set of buffers (rounds 1-2's step): from the second launch on "
                                                       "the 256-MB Infinity Cache holds part of what a launch reads and overwrites — not an HBM figure")
    tp = r0.get("traffic_pass")
    mc = r0.get("mix_ceiling")
    if tp or mc:
        ceil = {}
        if tp:
            ceil["traffic_pass_us"] = tp["rotating_us"]
            ceil["traffic_pass_gbs"] = algo_bytes / (tp["rotating_us"] * 1e-6) / 1e9
            ceil["kernel_over_best_probe"] = kern_ms * 1e3 / tp["rotating_us"]
            ceil["what"] = ("hrx_traffic_pass_device over the SAME buffer sets in the same rotation, replayed as the same kind of graph: the bytes of one launch — "
                            "same addresses, same 16-byte-per-lane instructions, same store policy, 4 reader + 4 writer waves per CU — with no DFA work"
                            " (string-major lines: hrx_traffic_pass_device_layout, the walker/storer kernel's 128-byte lines of eight strings per store instruction)")
            if "one_set_us" in tp:
                ceil["one_set_us"] = tp["one_set_us"]
                if r0.get("one_buffer_set"):
                    ceil["one_set_kernel_over_probe"] = r0["one_buffer_set"]["ms_per_step_median"] * 1e3 / tp["one_set_us"]
        if mc:
            ceil["mixceil"] = {"us_per_launch": mc, "what": "tools/mixceil --brief, a separate process with its own plain allocations: copy = plain dwordx4 copy of the "
                               "byte count; pair* = the bench line's slabs with write-back / streaming / mixed stores, one buffer set; *_fresh = over 8 buffer sets in turn"}
        line["roofline"]["mix_ceiling"] = ceil
    if r0.get("from_string_major_input"):
        f = dict(r0["from_string_major_input"])
        if "transpose_plus_launch_ms" in f:      # (--probes)
            f["frac_transpose_plus_launch"] = algo_bytes / (f["transpose_plus_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        f["frac_string_major_input_launch"] = algo_bytes / (f["string_major_input_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        f["what"] = ("the same K steps over the same buffer sets starting from the reference's input shape, B contiguous strings (lib.rs:311-315): transpose = "
                     "hrx_chars_to_position_major_device alone; transpose_plus_launch = that followed by the headline's launch (the cost of the fast path for a caller "
                     "holding &[u8] strings); string_major_input_launch = the position-major kernel reading the strings directly (each lane's 16-byte pieces a stride apart). "
                     "frac_* = the headline's algorithmic bytes over that time, of 8 TB/s")
        line["roofline"]["from_string_major_input"] = f
    if r0.get("cpu_baseline"):
        line["cpu_baseline"] = r0["cpu_baseline"]
    if r0.get("single_string"):
        line["single_string"] = r0["single_string"]
    if r0.get("end_to_end_host"):
        line["end_to_end_host"] = r0["end_to_end_host"]
    if r0.get("warmup_effective"):
        line["warmup_effective"] = r0["warmup_effective"]
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


def physical_device(dev_index):
    """HIP_VISIBLE_DEVICES value that selects logical device `dev_index` of THIS process in a child (the parent may itself be restricted)"""
    vis = os.environ.get("HIP_VISIBLE_DEVICES")
    if vis:
        ids = [v.strip() for v in vis.split(",") if v.strip()]
        if dev_index < len(ids):
            return ids[dev_index]
    return str(dev_index)


def spawn_children(args, argv, timeout_s=3600):
    """Bare `python bench.py --gpus N`: N fresh child processes, one per device.  This (parent) process never touches a GPU.  The children
    are polled: the first one that fails (or the overall timeout) takes its siblings down instead of leaving them in a rendezvous."""
    import tempfile
    devs = os.environ.get("HRX_BENCH_DEVICES")
    devices = [int(x) for x in devs.split(",")] if devs else list(range(args.gpus))
    procs, outs = [], []
    for attempt in range(3):      # (free_port() closes its socket before the children bind: retry on the rare lost race)
        port = free_port()
        procs, outs = [], []
        for r in range(args.gpus):
            env = dict(os.environ, HRX_BENCH_RANK=str(r), HRX_BENCH_WORLD=str(args.gpus), HRX_BENCH_PORT=str(port),
                       HRX_BENCH_DEVICE=str(devices[r % len(devices)]))
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(k, None)
            f = tempfile.TemporaryFile(mode="w+")
            outs.append(f)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv + ["--child"], env=env, stdout=f, text=True))
        deadline, failed = time.time() + timeout_s, None
        while any(p.poll() is None for p in procs):
            bad = [r for r, p in enumerate(procs) if p.poll() not in (None, 0)]
            if bad or time.time() > deadline:
                failed = bad[0] if bad else -1
                for p in procs:
                    if p.poll() is None:
                        p.kill()
                for p in procs:
                    p.wait()
                break
            time.sleep(0.05)
        if failed is None and all(p.returncode == 0 for p in procs):
            break
        codes = [p.returncode for p in procs]
        if failed == -1:
            sys.stderr.write("bench.py: timed out after %d s; children killed\n" % timeout_s)
            raise SystemExit(1)
        if attempt < 2 and any(c == 75 for c in codes):     # EX_TEMPFAIL: the rendezvous port was taken
            continue
        sys.stderr.write("bench.py: rank exit codes %s\n" % codes)
        raise SystemExit(1)
    results = []
    for r, f in enumerate(outs):
        f.seek(0)
        lines = [l for l in f.read().splitlines() if l.startswith(RESULT_TAG)]
        f.close()
        if not lines:
            sys.stderr.write("bench.py: rank %d reported nothing\n" % r)
            raise SystemExit(1)
        results.append(json.loads(lines[-1][len(RESULT_TAG):]))
    return aggregate(results, args)


# The BASELINE configs other than the one the metric is quoted on, at their per-GPU sizes: each runs as a child of the default single-GPU run AFTER the headline's
# timed region, verification and probes, through the same code path (rotating buffer sets, the K steps replayed as one HIP graph, every string of every
# set compared with the oracle, hrx_traffic_pass_device over the same buffers), and is condensed into the line's `other_configs`.
OTHER_CONFIGS = [
    ("configs[2]: regex2_test + regex3_test with substr extraction, 2^20 x 2048-byte strings, 1 MI355X",
     ["--config", "regex23", "--batch", "1048576", "--len", "2047", "--rows", "2048", "--steps", "10", "--warmup", "3", "--distinct", "65536", "--planes"]),
    ("configs[3]: 32-KiB header regexes (D = 3 stand-ins, BASELINE.md), 256K strings over 8 GPUs = 32768 strings per GPU",
     ["--config", "headers3", "--batch", "32768", "--len", "32767", "--rows", "32768", "--steps", "10", "--warmup", "3", "--distinct", "4096", "--planes"]),
    ("configs[4]: synthetic 256-state dense DFA, 4096-byte inputs, >= 1M strings over 8 GPUs = 131072 strings per GPU",
     ["--config", "dfa256", "--batch", "131072", "--len", "4095", "--rows", "4096", "--steps", "40", "--warmup", "8", "--distinct", "65536", "--planes"]),
]


def is_default_workload(args):
    return (args.config == "regex1" and args.batch == 65536 and args.n == 1023 and args.rows == 1024 and args.layout == "position-major" and args.gpus == 1
            and args.scaling == "weak" and args.dist == "planted" and not args.eager and not args.leg and args.distinct == 0)


def condense_leg(name, line, wall_s):
    r = line["roofline"]
    mc = r.get("mix_ceiling") or {}
    pl = r.get("placement") or {}
    v = line.get("verified") or {}
    return {"baseline_config": name, "workload": line["config"]["workload"], "value": line["value"], "unit": line["unit"], "steps": line["steps"], "warmup": line["warmup"],
            "ms_per_step": line["ms_per_step"], "avg_launch_ms": r["avg_launch_ms"], "frac": r["frac"], "achieved": r["achieved"], "peak": r["peak"],
            "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"], "bytes_per_row": r["bytes_per_row"], "kernel": r["kernel"], "launch": r["launch"],
            "traffic_pass_us": mc.get("traffic_pass_us"), "kernel_over_traffic_pass": mc.get("kernel_over_best_probe"),
            "spread_ms_per_step": (line.get("spread") or {}).get("ms_per_step_median"),
            "verified": {"bit_exact": v.get("bit_exact"), "strings": v.get("strings"), "buffer_sets": v.get("buffer_sets"), "distinct_strings": v.get("distinct_strings")},
            "placement": {"best_gbs": pl.get("best_gbs"), "ref_gbs": pl.get("ref_gbs"), "steps": pl.get("steps"), "sets_accepted": pl.get("sets_accepted"), "capped": pl.get("capped"),
                          "search_ms": pl.get("search_ms"), "busiest_class_bytes_per_row": pl.get("chosen_step") if "RECORD PLANES" in line["config"]["buffers"] else None},
            "record_planes": "RECORD PLANES" in line["config"]["buffers"], "per_set_ms": r.get("per_set_ms"),
            "interleaved_layout": ({k: r["interleaved_layout"].get(k) for k in ("ms_per_step_median", "frac", "kernel")} if r.get("interleaved_layout") else None),
            "buffer_sets": line["config"]["buffer_sets"].split(":")[0], "launch_mode": line["config"]["launch_mode"].split(" (")[0], "wall_s": wall_s}


def other_config_legs(dev_index, timeout_s=600):
    """Runs OTHER_CONFIGS one after the other, each in a fresh child process on the same device (this process keeps its context: the children are started,
    not exec'ed into).  A leg that fails, times out or does not fit the free device memory is reported as skipped with the reason; it never breaks the line."""
    legs = []
    if under_profiler():
        return [{"baseline_config": name, "skipped": "under a profiler: no child process may be started"} for name, _ in OTHER_CONFIGS]
    env = dict(os.environ, HIP_VISIBLE_DEVICES=physical_device(dev_index))
    for name, argv in OTHER_CONFIGS:
        t0 = time.time()
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv + ["--leg", "--no-cpu-baseline", "--no-pmc"], capture_output=True, text=True,
                               timeout=timeout_s, env=env)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not lines:
                legs.append({"baseline_config": name, "skipped": "leg failed (rc %d): %s" % (r.returncode, (r.stderr or "").strip().splitlines()[-1:] or "")})
                continue
            legs.append(condense_leg(name, json.loads(lines[-1]), time.time() - t0))
        except subprocess.TimeoutExpired:
            legs.append({"baseline_config": name, "skipped": "timed out after %d s" % timeout_s})
        except Exception as e:
            legs.append({"baseline_config": name, "skipped": "leg failed: %s" % e})
    return legs


def emit(line):
    """The ONE JSON line on stdout — with the headline's `roofline` block as its LAST key, so that a truncated tail of the output still shows it — and a one-line summary on stderr."""
    if "roofline" in line:
        line["roofline"] = line.pop("roofline")
    print(json.dumps(line), flush=True)
    try:
        r = line.get("roofline") or {}
        legs = " ".join("%s=%.3f" % (o["baseline_config"].split(":")[0], o["frac"]) for o in line.get("other_configs", []) if "frac" in o)
        sys.stderr.write("bench.py: %s = %.4g %s on %d GPU(s), %.4f ms/step; roofline %.3f of %d GB/s (%s, avg launch %.4f ms, traffic %s); verified %s; cpu_baseline %s %s\n" % (
            line.get("metric"), line.get("value", 0), line.get("unit"), line.get("n_gpus", 0), line.get("ms_per_step", 0), r.get("frac", 0), r.get("peak", 0), (r.get("kernel") or "")[:60],
            r.get("avg_launch_ms", 0), r.get("traffic"), (line.get("verified") or {}).get("bit_exact"), (line.get("cpu_baseline") or {}).get("value"), legs))
    except Exception:
        pass


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
            emit(line)
        dist.barrier()
        dist.destroy_process_group()
        return
    if args.child:
        # ---- one of the N children of a bare --gpus N run: gloo over 127.0.0.1 for the barrier, result over the pipe
        import torch.distributed as dist
        rank, world = int(os.environ["HRX_BENCH_RANK"]), int(os.environ["HRX_BENCH_WORLD"])
        import datetime
        try:
            dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["HRX_BENCH_PORT"], rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=120))
        except Exception as e:          # rank 0 could not bind the port the parent picked (taken in between): the parent retries with another
            sys.stderr.write("bench.py child %d: rendezvous failed: %s\n" % (rank, e))
            raise SystemExit(75)
        res = run_rank(args, rank, world, int(os.environ["HRX_BENCH_DEVICE"]), dist.barrier)
        print(RESULT_TAG + json.dumps(res), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    if args.gpus > 1:
        if under_profiler():   # the profiler's preloaded library has initialised the GPU in THIS process: it must not fork + exec children
            raise SystemExit("bench.py --gpus %d under rocprofv3: profile one rank (--gpus 1), or wrap each rank of a torch.distributed.run launch; "
                             "a bare multi-GPU run spawns children and must not start under a profiler" % args.gpus)
        print(json.dumps(spawn_children(args, argv)), flush=True)
        return
    res = run_rank(args, 0, 1, 0, lambda: None)
    line = aggregate([res], args)
    if is_default_workload(args) and not args.no_other_configs:
        line["other_configs"] = other_config_legs(0)
    emit(line)


if __name__ == "__main__":
    main()
