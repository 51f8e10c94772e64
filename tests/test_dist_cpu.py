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
import os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import halo2_regex_amd as hra
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
B = 1000
b, c = hra.shard_range(B, world, rank)
dist.barrier()
elapsed = torch.tensor([0.5 + 0.25 * rank], dtype=torch.float64)     # rank 1 is the slow one
dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
rows = torch.tensor([c * 1023], dtype=torch.int64)
dist.all_reduce(rows)
if rank == 0:
    print("RESULT", float(elapsed.item()), int(rows.item()), b, c)
dist.barrier()
dist.destroy_process_group()
"""


def test_world_size_2_reduction_on_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", str(script), ROOT],
                         capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][0].split()
    assert float(line[1]) == 0.75           # max over ranks
    assert int(line[2]) == 1000 * 1023      # whole-job rows: every string exactly once
    assert (int(line[3]), int(line[4])) == (0, 500)


def test_cpp_host_mirror_builds_and_runs():
    exe = "/tmp/hrx_test_host"
    csrc = os.path.join(ROOT, "halo2_regex_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "host_cpp", "test_host.cpp"), "-o", exe,
                           "-L" + csrc, "-lhrx", "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"])
    out = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "dfa")], capture_output=True, text=True)
    assert out.returncode == 0 and "host ok" in out.stdout
