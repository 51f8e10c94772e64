"""Pins the CPU oracle against every known answer the reference's own tests hold for the path.

Cases: tests/golden/reference_tests.json (extracted from /root/reference/src/lib.rs:1067-1470 and
examples/regex.rs:185-199 by tests/golden/extract_from_reference.py).
"""
import numpy as np
import pytest

from oracle_lib import OracleDefs, reference_cases, ORC_OK

CASES = reference_cases()


def _expected_columns(case):
    M = case["max_chars_size"]
    chars = np.zeros(M, np.uint64)
    ids = np.zeros(M, np.uint64)
    # lib.rs:1046-1051: substr_idx+1 is the expected substr id, in list order
    for k, (start, text) in enumerate(case["expected_substrs"]):
        for i, ch in enumerate(text.encode("latin-1")):
            chars[start + i] = ch
            ids[start + i] = k + 1
    return chars, ids


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_reference_known_answers(oracle, case):
    defs = OracleDefs.from_files(oracle, case["defs"])
    M = case["max_chars_size"]
    inp = case["input"].encode("latin-1")
    out = defs.match_substrs(inp, M)
    assert out["rc"] == ORC_OK
    if case["masked_outputs_asserted"]:
        exp_c, exp_i = _expected_columns(case)
        # lib.rs:1052-1059 / 1300-1307: masked_characters and all_substr_ids row by row
        assert np.array_equal(out["masked_char"], exp_c)
        assert np.array_equal(out["masked_substr_id"], exp_i)
    # MockProver::verify() == Ok  <=>  (for these witnesses) the state at the first padded row is the
    # accepted state for every def (gate chain lib.rs:427-457); the fail cases must not be accepted.
    n = len(inp)
    accepted = all(int(out["state"][d, n]) == int(oracle.orc_accepted_state(defs.h, d)) for d in range(defs.D))
    assert accepted == case["verify_ok"]
    acc_mask = int(out["info"][4])
    assert (acc_mask == (1 << defs.D) - 1) == case["verify_ok"]


def test_survey_appendix_c_traces(oracle):
    """Per-row traces listed in SURVEY.md App. C (states / nonzero sids / flags)."""
    A = OracleDefs.from_files(oracle, [["regex1_test_lookup.txt", ["substr1_test_lookup.txt"]],
                                       ["regex2_test_lookup.txt", ["substr2_test_lookup.txt"]]])
    s = A.derive_states(b"email was meant for @y. Also for x.")
    assert list(s[0]) == list(range(0, 23)) + [24] * 13
    assert list(s[1]) == [0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8,
                          9, 10, 11, 12]
    ids = A.derive_substr_ids(s)
    assert list(np.nonzero(ids[0])[0]) == [21] and ids[0][21] == 1
    assert list(np.nonzero(ids[1])[0]) == [33] and ids[1][33] == 2     # offset rule lib.rs:842
    st, en = A.derive_is_start_end(s, ids)
    assert list(np.nonzero(st[0])[0]) == [21] and list(np.nonzero(en[0])[0]) == [22]
    assert list(np.nonzero(st[1])[0]) == [33] and list(np.nonzero(en[1])[0]) == [34]
    assert st.shape == (2, 36) and en.shape == (2, 36)                 # n+1 each, lib.rs:869,882

    s = A.derive_states(b"email was meant for @yajk. Also for swq.")
    ids = A.derive_substr_ids(s)
    st, en = A.derive_is_start_end(s, ids)
    assert list(s[0][21:27]) == [21, 22, 22, 22, 22, 24]
    assert list(np.nonzero(ids[0])[0]) == [21, 22, 23, 24]
    assert list(np.nonzero(en[0])[0]) == [22, 23, 24, 25]
    assert list(np.nonzero(ids[1])[0]) == [36, 37, 38] and list(np.nonzero(en[1])[0]) == [37, 38, 39]

    Bd = OracleDefs.from_files(oracle, [["regex3_test_lookup.txt", ["substr3_test_lookup.txt"]]])
    s = Bd.derive_states(b"from:alice@gmail.com\r\n")
    assert list(s[0]) == [0, 6, 8, 10, 11, 12, 14, 14, 14, 14, 14, 17, 4, 4, 4, 4, 4, 4, 4, 4, 4, 18, 5]
    s = Bd.derive_states(b"dummy\r\nfrom:alice<alice@gmail.com>\r\n")
    assert list(s[0]) == [0, 1, 1, 1, 1, 1, 7, 9, 6, 8, 10, 11, 12, 14, 14, 14, 14, 14, 15, 3, 3, 3, 3, 3, 16, 2, 2, 2,
                          2, 2, 2, 2, 2, 2, 19, 18, 5]
    ids = Bd.derive_substr_ids(s)
    assert list(np.nonzero(ids[0])[0]) == list(range(12, 17)) + list(range(18, 33))
    st, en = Bd.derive_is_start_end(s, ids)
    assert list(np.nonzero(st[0])[0]) == [12, 18] and list(np.nonzero(en[0])[0]) == list(range(25, 34))
    # fail3: masks are non-zero even though the string is not accepted
    out = Bd.match_substrs(b"from:alice<alice@gmail.com>", 1024)
    assert out["rc"] == ORC_OK and out["masked_char"].any()


def test_invalid_transition_panics_like_reference(oracle):
    # examples/ex_allstr.txt is a partial DFA: state 2 (accept) has no out-edges (SURVEY App. C last row)
    ex = OracleDefs.from_files(oracle, [["ex_allstr.txt", ["ex_substr_id1.txt"]]])
    with pytest.raises(RuntimeError, match=r"The transition from 2 by 33 is invalid!"):   # lib.rs:817
        ex.derive_states(b"email was meant for @vitalik.!")
    out = ex.match_substrs(b"email was meant for @vitalik.!", 128)
    assert out["rc"] == 1 and list(out["info"][:4]) == [0, 29, 2, 33]
    # byte outside the alphabet of a total-on-alphabet DFA
    r1 = OracleDefs.from_files(oracle, [["regex1_test_lookup.txt", ["substr1_test_lookup.txt"]]])
    with pytest.raises(RuntimeError, match=r"The transition from 0 by 200 is invalid!"):
        r1.derive_states(bytes([200]))


def test_padding_rules(oracle):
    """state column holds s[n] at row n, then dummy = largest+1; n == M drops s[n]."""
    r1 = OracleDefs.from_files(oracle, [["regex1_test_lookup.txt", ["substr1_test_lookup.txt"]]])
    inp = b"email was meant for @y."
    n = len(inp)
    out = r1.match_substrs(inp, 64)
    assert list(out["enable"]) == [1] * n + [0] * (64 - n)
    assert int(out["state"][0, n]) == 24 and set(out["state"][0, n + 1:]) == {29}      # lib.rs:406-414
    assert not out["substr_id"][0, n:].any() and not out["start_enable"][0, n:].any()
    out = r1.match_substrs(inp, n)            # n == M: legal for the integers
    assert out["rc"] == ORC_OK and len(out["state"][0]) == n and int(out["state"][0, n - 1]) == 22
    assert list(out["masked_char"][21:23]) == [ord("y"), 0]
    out = r1.match_substrs(inp, n - 1)        # n > M: out of contract
    assert out["rc"] == 3


def test_table_rows_match_table_rs(oracle):
    """RegexTableConfig::load (table.rs:61-198): row 0 dummy, rows in file-line order, endpoint rows."""
    A = OracleDefs.from_files(oracle, [["regex1_test_lookup.txt", ["substr1_test_lookup.txt"]],
                                       ["regex2_test_lookup.txt", ["substr2_test_lookup.txt"]]])
    import os
    from oracle_lib import DFA_DIR
    for d, (fn, dummy, off) in enumerate([("regex1_test_lookup.txt", 29, 1), ("regex2_test_lookup.txt", 13, 2)]):
        rows = A.table_transition_rows(d)
        lines = open(os.path.join(DFA_DIR, fn)).read().split("\n")[3:]
        lines = [l for l in lines if l.strip()]
        assert len(rows) == 1 + len(lines)
        assert list(rows[0]) == [0, dummy, dummy, 0]
        for r, l in zip(rows[1:], lines):
            cur, nxt, ch = map(int, l.split())
            assert (int(r[0]), int(r[1]), int(r[2])) == (ch, cur, nxt)
            assert int(r[3]) in (0, off)
        tagged = {(int(r[1]), int(r[2])) for r in rows[1:] if r[3]}
        sub = open(os.path.join(DFA_DIR, "substr%d_test_lookup.txt" % (d + 1))).read().split("\n")[5:]
        assert tagged == {tuple(map(int, l.split())) for l in sub if l.strip()}
    ep = A.table_endpoint_rows(0)
    assert [list(map(int, r)) for r in ep] == [[0, 29, 29], [1, 21, 29]] + [[1, 29, e] for e in (22, 23, 25, 26, 27, 28)]
    ep = A.table_endpoint_rows(1)
    assert [list(map(int, r)) for r in ep] == [[0, 13, 13], [2, 10, 13], [2, 13, 11]]


def test_threaded_and_dense_cpu_variants_equal_the_port(oracle):
    """bench.py's extra CPU lines (all host cores; dense-table "best CPU") produce the port's bytes."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from halo2_regex_amd import synth
    cases = [([["regex1_test_lookup.txt", ["substr1_test_lookup.txt"]]], synth.ragged(300, 200, seed=4), 200),
             ([["regex1_test_lookup.txt", ["substr1_test_lookup.txt"]]], synth.reveal_stress(200, 256, seed=9), 256),
             ([["regex2_test_lookup.txt", ["substr2_test_lookup.txt"]], ["regex3_test_lookup.txt", ["substr3_test_lookup.txt"]]],
              synth.regex23_planted(200, 511, seed=1), 512),
             ([["regex2_test_lookup.txt", ["substr2_test_lookup.txt"]], ["regex3_test_lookup.txt", ["substr3_test_lookup.txt"]]],
              synth.ragged(200, 128, seed=8), 128)]
    for names, (chars, lens), M in cases:
        o = OracleDefs.from_files(oracle, names)
        rec, msk, st = o.witness_batch(chars, lens, M)
        ok = (st & np.uint64(0xff)) == 0
        assert ok.any()
        for kw in (dict(threads=4), dict(dense=True), dict(dense=True, threads=3)):
            r2, m2, s2 = o.witness_batch(chars, lens, M, **kw)
            assert np.array_equal(st, s2), kw
            assert np.array_equal(rec[ok], r2[ok]) and np.array_equal(msk[ok], m2[ok]), kw
