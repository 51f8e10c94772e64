"""Algorithm check without a GPU: the per-lane tile algebra the kernel runs (csrc/hrx_lane.h) and the dense
fused tables (csrc/hrx_defs.cpp), re-enacted lane by lane on the CPU by tests/sim/lane_sim.cpp, against the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle_lib import OracleDefs, DFA_DIR, ROOT, reference_cases
from halo2_regex_amd import synth

SIM_SRC = os.path.join(ROOT, "tests", "sim", "lane_sim.cpp")
SIM_SO = os.path.join(ROOT, "tests", "sim", "_build", "liblane_sim.so")
CSRC = os.path.join(ROOT, "halo2_regex_amd", "csrc")


@pytest.fixture(scope="module")
def sim():
    deps = [SIM_SRC, os.path.join(CSRC, "hrx_defs.cpp"), os.path.join(CSRC, "hrx_lane.h"), os.path.join(CSRC, "hrx_defs.hpp")]
    if not os.path.exists(SIM_SO) or any(os.path.getmtime(SIM_SO) < os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(SIM_SO), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", SIM_SO, SIM_SRC,
                               os.path.join(CSRC, "hrx_defs.cpp")])
    lib = C.CDLL(SIM_SO)
    lib.sim_new.restype = C.c_void_p
    lib.sim_free.argtypes = [C.c_void_p]
    lib.sim_push_allstr.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.sim_push_substr.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.sim_finalize.argtypes = [C.c_void_p]
    lib.sim_check_half_image.argtypes = [C.c_void_p]
    lib.sim_check_half_image.restype = C.c_long
    lib.sim_check_byte_image.argtypes = [C.c_void_p]
    lib.sim_check_byte_image.restype = C.c_long
    lib.sim_witness_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.sim_witness_batch_w.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    for f in (lib.sim_fill_up, lib.sim_fill_down):
        f.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
        f.restype = C.c_uint64
    return lib


class SimDefs:
    def __init__(self, lib, defs):
        self.lib = lib
        self.h = lib.sim_new()
        for allstr, substrs in defs:
            t = open(os.path.join(DFA_DIR, allstr), "rb").read() if isinstance(allstr, str) else allstr
            assert lib.sim_push_allstr(self.h, t, len(t)) == 0
            for s in substrs:
                t = open(os.path.join(DFA_DIR, s), "rb").read() if isinstance(s, str) else s
                assert lib.sim_push_substr(self.h, t, len(t)) == 0
        assert lib.sim_finalize(self.h) == 0
        self.D = len(defs)

    def run(self, chars, lens, M, W=64):
        chars = np.ascontiguousarray(chars, np.uint8)
        lens = np.ascontiguousarray(lens, np.uint32)
        B, stride = chars.shape
        rec = np.zeros((B, M, self.D), np.uint32)
        msk = np.zeros((B, M), np.uint16)
        st = np.zeros(B, np.uint64)
        fix = np.zeros(1, np.uint64)
        self.lib.sim_witness_batch_w(self.h, W, chars.ctypes.data, stride, lens.ctypes.data, B, M, rec.ctypes.data,
                                     msk.ctypes.data, st.ctypes.data, fix.ctypes.data)
        return rec, msk, st, int(fix[0])


CFG_A = [["regex1_test_lookup.txt", ["substr1_test_lookup.txt"]], ["regex2_test_lookup.txt", ["substr2_test_lookup.txt"]]]
CFG_1 = [["regex1_test_lookup.txt", ["substr1_test_lookup.txt"]]]
CFG_23 = [["regex2_test_lookup.txt", ["substr2_test_lookup.txt"]], ["regex3_test_lookup.txt", ["substr3_test_lookup.txt"]]]
CFG_3 = [["regex3_test_lookup.txt", ["substr3_test_lookup.txt"]]]
CFG_EX = [["ex_allstr.txt", ["ex_substr_id1.txt"]]]


def _compare(oracle, sim, cfg, chars, lens, M, widths=(64, 32, 16)):
    """every tile width the kernels use: 64 (one-wave kernel), 32 and 16 (walker/storer kernel)"""
    o = OracleDefs.from_files(oracle, cfg)
    s = SimDefs(sim, cfg)
    orec, omsk, ost = o.witness_batch(chars, lens, M)
    ok = (ost & np.uint64(0xff)) == 0
    fixes = []
    for W in widths:
        srec, smsk, sst, fix = s.run(chars, lens, M, W)
        assert np.array_equal(ost, sst), W
        assert np.array_equal(orec[ok], srec[ok]), W
        assert np.array_equal(omsk[ok], smsk[ok]), W
        fixes.append(fix)
    return ost, omsk, fixes[0]


def test_scan_primitives_match_the_sequential_recurrence(sim):
    rng = np.random.default_rng(0)
    for _ in range(2000):
        dens = rng.choice([2, 4, 16])
        a = int(rng.integers(0, 2**63)) & int(rng.integers(0, 2**63)) if dens > 2 else int(rng.integers(0, 2**63))
        b = int(rng.integers(0, 2**63))
        setm = a & ~b & (2**64 - 1)
        rstm = b & ~a & (2**64 - 1)
        if dens == 16:
            setm &= int(rng.integers(0, 2**63)); rstm &= int(rng.integers(0, 2**63))
        for cin in (0, 1):
            last, up = cin, 0
            for i in range(64):          # lib.rs:631-642
                if (setm >> i) & 1: last = 1
                if (rstm >> i) & 1: last = 0
                up |= last << i
            last, dn = cin, 0
            for i in range(63, -1, -1):  # lib.rs:699-710
                if (setm >> i) & 1: last = 1
                if (rstm >> i) & 1: last = 0
                dn |= last << i
            assert sim.sim_fill_up(setm, rstm, cin) == up
            assert sim.sim_fill_down(setm, rstm, cin) == dn


@pytest.mark.parametrize("case", reference_cases(), ids=[c["name"] for c in reference_cases()])
def test_reference_cases(oracle, sim, case):
    inp = case["input"].encode("latin-1")
    M = case["max_chars_size"]
    chars = np.zeros((1, (len(inp) + 15) // 16 * 16), np.uint8)
    chars[0, :len(inp)] = np.frombuffer(inp, np.uint8)
    _compare(oracle, sim, case["defs"], chars, np.array([len(inp)], np.uint32), M)


@pytest.mark.parametrize("M", [1, 7, 63, 64, 65, 128, 200, 1024])
def test_ragged_lengths_and_row_counts(oracle, sim, M):
    chars, lens = synth.ragged(96, M, seed=M)
    st, _, _ = _compare(oracle, sim, CFG_1, chars, lens, M)
    assert ((st & np.uint64(0xff)) == 0).all()
    _compare(oracle, sim, CFG_A, chars, lens, M)


def test_reveal_mask_stress_crosses_tiles_and_needs_fixups(oracle, sim):
    chars, lens = synth.reveal_stress(600, 700, seed=11)
    for cfg in (CFG_1, CFG_3, CFG_A, CFG_23):
        st, msk, fix = _compare(oracle, sim, cfg, chars, lens, 704)
    # the stress set must actually exercise the optimistic end-mask protocol, both ways
    _, msk, fix = _compare(oracle, sim, CFG_3, chars, lens, 704)
    assert fix > 0 and msk.any()


def test_invalid_transition_and_bad_length_status(oracle, sim):
    chars, lens = synth.ragged(64, 300, seed=3)
    rng = np.random.default_rng(1)
    for b in range(0, 64, 3):                      # bytes outside the alphabet -> lib.rs:817
        if lens[b]:
            chars[b, int(rng.integers(0, lens[b]))] = 200 + b % 50
    lens[5] = 400                                  # n > M
    st, _, _ = _compare(oracle, sim, CFG_A, chars, lens, 304)
    codes = st & np.uint64(0xff)
    assert (codes == 1).any() and (codes == 3).any() and (codes == 0).any()
    # partial DFA of the example: state 2 has no out-edges
    inp = b"email was meant for @vitalik.!"
    c = np.zeros((1, 32), np.uint8); c[0, :len(inp)] = np.frombuffer(inp, np.uint8)
    st, _, _ = _compare(oracle, sim, CFG_EX, c, np.array([len(inp)], np.uint32), 128)
    assert int(st[0]) & 0xff == 1 and int(st[0]) >> 40 == 29


def test_flag_overlap_is_reported(oracle, sim):
    # the same def twice: both defs raise start/end flags on the same rows -> out of contract
    cfg = [CFG_1[0], CFG_1[0]]
    inp = b"email was meant for @ab."
    c = np.zeros((2, 32), np.uint8); c[0, :len(inp)] = np.frombuffer(inp, np.uint8)
    st, _, _ = _compare(oracle, sim, cfg, c, np.array([len(inp), 0], np.uint32), 64)
    assert int(st[0]) & 0xff == 2 and int(st[1]) & 0xff == 0


def test_half_table_image_equals_the_fused_table(sim):
    """The 2-byte HALF image (LDS-resident tables for DFAs of up to 256 states, cfg 5) carries exactly the transitions,
    substr ids and start/end flags of the 4-byte fused table; it does not exist beyond 256 states."""
    for cfg, states in ((CFG_1, 29), (CFG_A, 29 + 13), (CFG_23, 13 + 20)):
        assert sim.sim_check_half_image(SimDefs(sim, cfg).h) == states * 256
    allb = np.arange(256, dtype=np.uint8)
    a_txt, sub_txt = synth.random_dfa(256, seed=2, alphabet=allb, n_substr_pairs=200)
    assert sim.sim_check_half_image(SimDefs(sim, [(a_txt.encode(), [sub_txt.encode()])]).h) == 256 * 256
    a_txt, sub_txt = synth.random_dfa(300, seed=5, alphabet=allb[:64], n_substr_pairs=50)
    assert sim.sim_check_half_image(SimDefs(sim, [(a_txt.encode(), [sub_txt.encode()])]).h) == -1


def test_byte_table_image_equals_the_fused_table(sim):
    """The BYTE image (1-byte next-state table on the chain + the (state, next) pair tags in a perfect-hash table off it; one
    def of at most 256 table rows, cfg 5) carries exactly the transitions, substr ids and start/end flags of the 4-byte fused
    table: total DFAs have no dead row, partial ones one more row than states; two defs or more than 256 rows: no image."""
    for cfg, rows in ((CFG_1, 29 + 1), (CFG_3, 20 + 1)):                        # (98-symbol alphabets: partial over the 256 byte values -> + the dead row)
        assert sim.sim_check_byte_image(SimDefs(sim, cfg).h) == rows * 256
    allb = np.arange(256, dtype=np.uint8)
    a_txt, sub_txt = synth.random_dfa(256, seed=2, alphabet=allb, n_substr_pairs=200)
    assert sim.sim_check_byte_image(SimDefs(sim, [(a_txt.encode(), [sub_txt.encode()])]).h) == 256 * 256          # total: 256 rows
    a_txt, sub_txt = synth.random_dfa(255, seed=3, total=False, alphabet=allb, n_substr_pairs=250)
    assert sim.sim_check_byte_image(SimDefs(sim, [(a_txt.encode(), [sub_txt.encode()])]).h) == 256 * 256          # partial: 255 states + the dead row
    a_txt, sub_txt = synth.random_dfa(255, seed=3, total=False, alphabet=allb, n_substr_pairs=900)
    assert sim.sim_check_byte_image(SimDefs(sim, [(a_txt.encode(), [sub_txt.encode()])]).h) == -1                 # too many tagged pairs for a collision-free hash: the HALF table serves it
    a_txt, sub_txt = synth.random_dfa(256, seed=4, total=False, alphabet=allb, n_substr_pairs=50)
    assert sim.sim_check_byte_image(SimDefs(sim, [(a_txt.encode(), [sub_txt.encode()])]).h) == -1                 # partial with 256 states: no row left for the dead state
    assert sim.sim_check_byte_image(SimDefs(sim, CFG_A).h) == -1                                                  # two defs
