"""The N > 1 path without GPUs: sharding arithmetic against the oracle, and bench.py's world-size-2 reduction logic
(barrier + max-over-ranks + aggregate) on the gloo backend."""
import os
import subprocess
import sys

import numpy as np

import halo2_regex_amd as hra
from halo2_regex_amd import synth
from oracle_lib import OracleDefs, ROOT

CFG_1 = [["regex1_test_lookup.txt", ["substr1_test_lookup.txt"]]]


def test_shards_reproduce_the_whole(oracle):
    """Strings are independent given the RegexDefs: the concatenation of per-rank results is the batch result."""
    chars, lens = synth.reveal_stress(203, 300, seed=5)
    o = OracleDefs.from_files(oracle, CFG_1)
    rec, msk, st = o.witness_batch(chars, lens, 304)
    for world in (2, 3, 8):
        parts = [hra.shard_range(len(lens), world, r) for r in range(world)]
        r2 = np.concatenate([o.witness_batch(chars[b:b + c], lens[b:b + c], 304)[0] for b, c in parts if c])
        m2 = np.concatenate([o.witness_batch(chars[b:b + c], lens[b:b + c], 304)[1] for b, c in parts if c])
        assert np.array_equal(rec, r2) and np.array_equal(msk, m2)


WORKER = r"""
# one rank of a world-size-2 job on gloo: bench.py's own gather + aggregate (the code `torch.distributed.run bench.py --gpus 2`
# executes after the timed region), fed with this rank's shard and a made-up elapsed time
import json, os, sys
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import bench
import halo2_regex_amd as hra
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
args = bench.parse_args(["--gpus", "2", "--steps", "7", "--warmup", "1", "--batch", "500"])
b, c = hra.shard_range(1000, world, rank)
res = {"rank": rank, "device": rank, "rows": c * 1023 * args.steps, "elapsed_s": 0.5 + 0.25 * rank, "avg_launch_ms": 0.08 + 0.01 * rank,
       "debug_flags": None}
if rank == 0:
    res.update(desc="hrx::witness_pm_kernel<1, false, false, false, false, false> grid=256 waves=8 ring=4 lds=113728", D=1, rows_per_step=c * 1023,
               config={"workload": "test"}, verified={"strings": 4, "rows": 4092, "bit_exact": True, "against": "test"})
dist.barrier()
line = bench.gather_and_aggregate(res, args)
assert (line is None) == (rank != 0)
if rank == 0:
    print("RESULT " + json.dumps(line))
dist.barrier()
dist.destroy_process_group()
"""


def test_world_size_2_reduction_on_gloo(tmp_path):
    """bench.py's gather_and_aggregate / aggregate under torch.distributed (gloo, world size 2): value = rows of ALL ranks /
    MAX over ranks of the elapsed time; per-rank results and ranks_seen are reported."""
    import json
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", str(script), ROOT],
                         capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["steps"] == 7
    assert line["value"] == 1000 * 1023 * 7 / 0.75          # every string exactly once, over the slow rank's time
    assert abs(line["ms_per_step"] - 750.0 / 7) < 1e-9
    assert [r["rows"] for r in line["per_rank"]] == [500 * 1023 * 7] * 2 and [r["device"] for r in line["per_rank"]] == [0, 1]
    assert line["verified"]["bit_exact"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert abs(line["roofline"]["achieved"] - 7 * 500 * 1023 / 0.08e-3 / 1e9) < 1e-6 and line["roofline"]["kernel"].startswith("hrx::witness_pm_kernel")


def test_aggregate_of_a_bare_multi_gpu_run():
    """the same aggregation as the parent of a bare `bench.py --gpus N` run applies it to its children's result lines"""
    import bench
    args = bench.parse_args(["--gpus", "4", "--steps", "5"])
    per = [{"rank": r, "device": r, "rows": 100 * 5, "elapsed_s": 1.0 + 0.1 * r, "avg_launch_ms": 0.1, "debug_flags": None} for r in (2, 0, 3, 1)]
    per[1].update(desc="k grid=1", D=2, rows_per_step=100, config={"workload": "t"})
    line = bench.aggregate(per, args)
    assert line["n_gpus"] == 4 and line["value"] == 2000 / 1.3 and [r["rank"] for r in line["per_rank"]] == [0, 1, 2, 3]
    assert line["roofline"]["bytes_per_row"] == 11 and line["roofline"]["algorithmic_bytes_per_launch"] == 1100


def test_cpp_host_mirror_builds_and_runs():
    exe = "/tmp/hrx_test_host"
    csrc = os.path.join(ROOT, "halo2_regex_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "host_cpp", "test_host.cpp"), "-o", exe,
                           "-L" + csrc, "-lhrx", "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"])
    out = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "dfa")], capture_output=True, text=True)
    assert out.returncode == 0 and "host ok" in out.stdout


def test_placement_walk_rule_on_recorded_candidate_sequences():
    """csrc/hrx_place_rule.hpp (when hrx_alloc_output_pair's walk over placement candidates stops) replayed on the probe rates round 4's leases recorded: a middle-kind
    candidate next to a slow one is not taken for a clear one, a later buffer set does not settle for less than an earlier one found, an arena walk passes its
    soft cap only while nothing is clear of the reference, every walk is bounded in time."""
    exe = "/tmp/hrx_test_place_rule"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "host_cpp", "test_place_rule.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "place rule: ok" in out.stdout, out.stdout + out.stderr


def test_arena_offset_allocator_on_a_host():
    """csrc/hrx_arena_alloc.hpp (the offsets inside a placement arena: first fit, freed ranges merge): eight bench-sized buffers back to back, a million-step alloc / free
    churn that never exhausts the arena, a seeded random trace against a byte map."""
    exe = "/tmp/hrx_test_arena_alloc"
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", os.path.join(ROOT, "tests", "host_cpp", "test_arena_alloc.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "arena ranges: ok" in out.stdout, out.stdout + out.stderr


def test_c_struct_entry_points_equal_the_text_parsers():
    """tests/host_c/test_push_structs.c replays bindings/rust/hrx.rs HrxHandle::new in C: hrx_defs_push_allstr with the map's entries SHUFFLED and their explicit
    line indices (table.rs:103-108), a duplicate key (defs.rs:100: the last insert wins), hrx_defs_push_substr with shuffled pairs — same fixed-table rows and the
    same witness rows (host walk here) as the text parsers."""
    exe = "/tmp/hrx_test_push_structs"
    csrc = os.path.join(ROOT, "halo2_regex_amd", "csrc")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "host_c", "test_push_structs.c"), "-o", exe,
                           "-L" + csrc, "-lhrx", "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"])
    out = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "dfa")], capture_output=True, text=True)
    assert out.returncode == 0 and "host ok" in out.stdout, out.stdout + out.stderr

