"""The integer MockProver (tests/mock_prover.py: the reference's gates, lookups and accept chain over plain integers,
src/lib.rs:126-305, 427-457; tables from src/table.rs:61-198) on the reference's own test inputs and on tampered witnesses.
CPU only: the witnesses come from the library's native host walk and from the oracle; the GPU parity tests run the same
checker over every full-size device output (tests/test_parity_gpu.py)."""
import os

import numpy as np
import pytest
import torch

import halo2_regex_amd as hra
from halo2_regex_amd import synth
from mock_prover import (IntegerMockProver, VERIFY_BITS, FAIL_ACCEPT, FAIL_TRANSITION, FAIL_FIRST_STATE, FAIL_START_LOOKUP,
                         FAIL_END_LOOKUP, FAIL_FLAGS, FAIL_PADDING, FAIL_MASK)
from oracle_lib import OracleDefs, DFA_DIR, reference_cases

CFG_1 = [["regex1_test_lookup.txt", ["substr1_test_lookup.txt"]]]
CFG_A = [["regex1_test_lookup.txt", ["substr1_test_lookup.txt"]], ["regex2_test_lookup.txt", ["substr2_test_lookup.txt"]]]
CFG_3 = [["regex3_test_lookup.txt", ["substr3_test_lookup.txt"]]]
CFG_123 = CFG_A + CFG_3


def _cfg(names, M):
    defs = [hra.RegexDefs(hra.AllstrRegexDef.read_from_text(os.path.join(DFA_DIR, a)),
                          [hra.SubstrRegexDef.read_from_text(os.path.join(DFA_DIR, s)) for s in subs]) for a, subs in names]
    return hra.RegexVerifyConfig.configure(M, defs, device=hra.HRX_DEVICE_NONE)


def _t(a):
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a.view({np.dtype(np.uint32): np.int32, np.dtype(np.uint16): np.int16}.get(a.dtype, a.dtype)))


def _verify(mp, chars, lens, rec, msk, M):
    return mp.verify(_t(chars), torch.from_numpy(np.asarray(lens).astype(np.int64)), _t(rec), _t(msk), M).numpy()


@pytest.mark.parametrize("case", reference_cases(), ids=[c["name"] for c in reference_cases()])
def test_reference_tests_verify_exactly_where_mockprover_does(oracle, case):
    """src/lib.rs:1067-1470, examples/regex.rs:185-199: `prover.verify()` is Ok for the pass cases and Err for every
    test_substr_fail* input.  Each fail input walks fine (all transitions defined) and ends in a non-accepting state, so the
    one constraint that breaks is the accept chain's assert_equal at the first disabled row (lib.rs:427-457)."""
    M = case["max_chars_size"]
    s = case["input"].encode("latin-1")
    chars = np.zeros((1, (len(s) + 15) // 16 * 16 or 16), np.uint8)
    chars[0, :len(s)] = np.frombuffer(s, np.uint8)
    lens = np.array([len(s)], np.uint32)
    cfg = _cfg(case["defs"], M)
    mp = IntegerMockProver.from_config(cfg)
    for rec, msk, st in (cfg.witness_batch_host(chars, lens), OracleDefs.from_files(oracle, case["defs"]).witness_batch(chars, lens, M)):
        code = int(_verify(mp, chars, lens, rec, msk, M)[0])
        if case["verify_ok"]:
            assert code == 0, mp.explain(code)
        else:
            assert code == FAIL_ACCEPT, mp.explain(code)      # verify() fails, and only through the accept chain


def test_every_tampered_cell_is_caught(oracle):
    """A witness that differs from the walk in one cell breaks the constraint that pins that cell."""
    M = 128
    s = b"email was meant for @yajk. Also for swq."
    chars = np.zeros((1, 48), np.uint8)
    chars[0, :len(s)] = np.frombuffer(s, np.uint8)
    lens = np.array([len(s)], np.uint32)
    cfg = _cfg(CFG_A, M)
    mp = IntegerMockProver.from_config(cfg)
    rec, msk, st = cfg.witness_batch_host(chars, lens)
    assert int(_verify(mp, chars, lens, rec, msk, M)[0]) == 0

    def tampered(fn):
        r, m = rec.copy(), msk.copy()
        fn(r, m)
        return int(_verify(mp, chars, lens, r, m, M)[0])

    def set_state(r, m): r[0, 10, 0] = (r[0, 10, 0] & np.uint32(0xffff0000)) | np.uint32(3)
    code = tampered(set_state)                                 # a wrong state: rows 9 and 10 leave the transition table
    assert code & FAIL_TRANSITION
    def first(r, m): r[0, 0, 1] = (r[0, 0, 1] & np.uint32(0xffff0000)) | np.uint32(1)
    assert tampered(first) & FAIL_FIRST_STATE
    def sid(r, m): r[0, 3, 0] |= np.uint32(1 << 16)           # a substr id the table does not hold for that transition
    assert tampered(sid) & FAIL_TRANSITION
    def start(r, m): r[0, 22, 0] |= np.uint32(1 << 24)       # start_enable on a row whose state is not a start state
    assert tampered(start) & FAIL_START_LOOKUP
    def end(r, m): r[0, 21, 0] |= np.uint32(1 << 25)         # row 21's next state (22) IS an end state of substr 1: the flag is already set
    assert tampered(end) == 0
    def end_bad(r, m): r[0, 20, 0] |= np.uint32(1 << 25)      # row 20: untagged (substr id 0) -> (0, dummy, state) is not an endpoint row
    assert tampered(end_bad) & FAIL_END_LOOKUP
    def drop_start(r, m): r[0, 21, 0] &= ~np.uint32(1 << 24)  # verify() cannot see a dropped flag; the completeness check does
    code = tampered(drop_start)
    assert code & FAIL_FLAGS and not code & (VERIFY_BITS & ~FAIL_START_LOOKUP)
    def pad_state(r, m): r[0, 100, 0] = 0                       # padding rows hold the dummy state
    assert tampered(pad_state) & FAIL_PADDING
    def accept(r, m): r[0, len(s), 0] = (r[0, len(s), 0] & np.uint32(0xffff0000)) | np.uint32(3)
    code = tampered(accept)
    assert code & FAIL_ACCEPT
    def mchar(r, m): m[0, 21] = 0
    assert tampered(mchar) == FAIL_MASK
    def mextra(r, m): m[0, 5] = 0x0161
    assert tampered(mextra) == FAIL_MASK


@pytest.mark.parametrize("names", [CFG_1, CFG_3, CFG_A, CFG_123], ids=["r1", "r3", "r1r2", "r1r2r3"])
def test_host_walk_and_oracle_witnesses_satisfy_the_constraints(oracle, names):
    """reveal_stress batches (spans that cross tiles, spans cut by the string's end, reset events) and ragged lengths incl.
    n = 0 and n = M: every row of every string passes; strings that do not reach the accept state fail only the accept chain."""
    M = 264
    cfg = _cfg(names, M)
    mp = IntegerMockProver.from_config(cfg)
    o = OracleDefs.from_files(oracle, names)
    for chars, lens in (synth.reveal_stress(300, 260, seed=5), synth.ragged(300, M, seed=7)):
        lens = lens.copy()
        lens[0], lens[1] = 0, min(M, chars.shape[1])
        for rec, msk, st in (cfg.witness_batch_host(chars, lens), o.witness_batch(chars, lens, M, threads=4)):
            code = _verify(mp, chars, lens, rec, msk, M)
            ok = (st & np.uint64(0xff)) == 0                   # (bytes outside the DFA's alphabet: the reference panics, no witness)
            accept_all = ((st >> np.uint64(8)) & np.uint64(0xff)) == np.uint64((1 << len(names)) - 1)
            full = lens >= M                                   # n == M: no row where enable drops, acceptance is not constrained (lib.rs:427-457)
            assert (code[ok & (accept_all | full)] == 0).all()
            rest = ok & ~accept_all & ~full
            assert (code[rest] == FAIL_ACCEPT).all()
            assert ok.sum() > 100


def test_random_definitions(oracle):
    """random DFAs and substring definitions: the table rows the checker uses come from the library, the witness from the host walk"""
    for seed in range(6):
        nstates = [5, 17, 40, 120, 256, 300][seed]
        allstr, sub = synth.random_dfa(nstates, seed=seed, total=True, n_substr_pairs=30)
        defs = [hra.RegexDefs(hra.AllstrRegexDef(allstr), [hra.SubstrRegexDef(sub)])]
        M = 200
        cfg = hra.RegexVerifyConfig.configure(M, defs, device=hra.HRX_DEVICE_NONE)
        mp = IntegerMockProver.from_config(cfg)
        chars, lens = synth.ragged(120, M, seed=seed, planted=False)
        rec, msk, st = cfg.witness_batch_host(chars, lens)
        code = _verify(mp, chars, lens, rec, msk, M)
        ok = (st & np.uint64(0xff)) == 0
        assert ok.sum() > 60
        assert ((code[ok] & ~FAIL_ACCEPT) == 0).all(), [mp.explain(c) for c in code[ok][:5]]
