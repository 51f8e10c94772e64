"""The library's native small-batch host walk (csrc/hrx_host_walk.cpp) against the oracle and against the reference's own
known answers — CPU only, through the C ABI on a host-only context (hrx_ctx_create(defs, HRX_DEVICE_NONE)).

This is the path the reference-shaped single-string surface takes (match_substrs hands over ONE string per call,
src/lib.rs:316-318).  It is the lane algorithm of csrc/hrx_lane.h (dense fused table, per-tile position bitvectors,
carry-chain mask scans with the optimistic end-mask protocol) on a host core; it neither links nor calls the oracle.
"""
import os

import numpy as np
import pytest

import halo2_regex_amd as hra
from halo2_regex_amd import synth
from oracle_lib import OracleDefs, DFA_DIR, reference_cases

CFG_1 = [["regex1_test_lookup.txt", ["substr1_test_lookup.txt"]]]
CFG_A = [["regex1_test_lookup.txt", ["substr1_test_lookup.txt"]], ["regex2_test_lookup.txt", ["substr2_test_lookup.txt"]]]
CFG_3 = [["regex3_test_lookup.txt", ["substr3_test_lookup.txt"]]]
CFG_23 = [["regex2_test_lookup.txt", ["substr2_test_lookup.txt"]], ["regex3_test_lookup.txt", ["substr3_test_lookup.txt"]]]
CFG_123 = CFG_A + CFG_3
CFG_EX = [["ex_allstr.txt", ["ex_substr_id1.txt"]]]


def _cfg(names, M):
    defs = [hra.RegexDefs(hra.AllstrRegexDef.read_from_text(os.path.join(DFA_DIR, a)),
                          [hra.SubstrRegexDef.read_from_text(os.path.join(DFA_DIR, s)) for s in subs]) for a, subs in names]
    return hra.RegexVerifyConfig.configure(M, defs, device=hra.HRX_DEVICE_NONE)


def _check_batch(oracle, names, chars, lens, M, cfg=None, o=None):
    cfg = cfg or _cfg(names, M)
    o = o or OracleDefs.from_files(oracle, names)
    orec, omsk, ost = o.witness_batch(chars, lens, M, threads=8)
    grec, gmsk, gst = cfg.witness_batch_host(chars, lens)
    assert np.array_equal(ost, gst)
    ok = (ost & np.uint64(0xff)) == 0
    assert np.array_equal(orec[ok], grec[ok])
    assert np.array_equal(omsk[ok], gmsk[ok])
    return ost, omsk


@pytest.mark.parametrize("case", reference_cases(), ids=[c["name"] for c in reference_cases()])
def test_reference_known_answers_on_the_host_walk(oracle, case):
    """src/lib.rs:1067-1470 + examples/regex.rs:185-199, through match_substrs of a host-only config."""
    M = case["max_chars_size"]
    cfg = _cfg(case["defs"], M)
    inp = case["input"].encode("latin-1")
    result = cfg.match_substrs(inp)                                   # lib.rs:1042
    expected_masked_chars = np.zeros(M, np.uint64)
    expected_substr_ids = np.zeros(M, np.uint64)
    for substr_idx, (start, chars) in enumerate(case["expected_substrs"]):   # lib.rs:1046-1051
        for idx, ch in enumerate(chars.encode("latin-1")):
            expected_masked_chars[start + idx] = ch
            expected_substr_ids[start + idx] = substr_idx + 1
    if case["masked_outputs_asserted"]:
        assert np.array_equal(result.masked_characters, expected_masked_chars)   # lib.rs:1052-1059
        assert np.array_equal(result.all_substr_ids, expected_substr_ids)
    accepted = hra.decode_status(result.status)["accept"] == (1 << cfg.num_defs) - 1
    assert accepted == case["verify_ok"]                              # MockProver::verify() outcome
    o = OracleDefs.from_files(oracle, case["defs"]).match_substrs(inp, M)
    for mine, theirs in (("all_enable_flags", "enable"), ("all_characters", "character"), ("states", "state"), ("substr_ids", "substr_id"),
                         ("start_enables", "start_enable"), ("end_enables", "end_enable"), ("masked_characters", "masked_char"),
                         ("all_substr_ids", "masked_substr_id")):
        assert np.array_equal(getattr(result, mine), o[theirs]), mine


@pytest.mark.parametrize("case", reference_cases()[:6], ids=[c["name"] for c in reference_cases()[:6]])
def test_derive_functions_match_lib_rs_804_888(oracle, case):
    cfg = _cfg(case["defs"], case["max_chars_size"])
    o = OracleDefs.from_files(oracle, case["defs"])
    inp = case["input"].encode("latin-1")
    states = cfg.derive_states(inp)
    assert states.shape == (cfg.num_defs, len(inp) + 1) and np.array_equal(states, o.derive_states(inp))
    sids = cfg.derive_substr_ids(states)
    assert np.array_equal(sids, o.derive_substr_ids(states))
    st, en = cfg.derive_is_start_end(states, sids)
    ost, oen = o.derive_is_start_end(states, sids)
    assert np.array_equal(st, ost) and np.array_equal(en, oen)


def test_invalid_transition_panics_with_the_reference_message():
    cfg = _cfg(CFG_EX, 128)
    with pytest.raises(hra.HrxError, match=r"^The transition from 2 by 33 is invalid!$") as e:   # lib.rs:817
        cfg.derive_states(b"email was meant for @vitalik.!")
    assert e.value.code == hra.HRX_ERR_INVALID_TRANSITION
    with pytest.raises(hra.HrxError, match=r"The transition from 0 by 200 is invalid!"):
        _cfg(CFG_1, 64).match_substrs(bytes([200]))
    assert _cfg(CFG_1, 64).derive_states(b"").tolist() == [[0]]
    with pytest.raises(hra.HrxError) as e:
        _cfg(CFG_1, 8).match_substrs(b"123456789")               # n > max_chars_size
    assert e.value.code == hra.HRX_ERR_OUT_OF_CONTRACT


@pytest.mark.parametrize("M", [1, 7, 8, 63, 64, 65, 72, 128, 200, 1024])
def test_ragged_batches_every_row_count(oracle, M):
    chars, lens = synth.ragged(200, M, seed=M)
    _check_batch(oracle, CFG_1, chars, lens, M)
    _check_batch(oracle, CFG_A, chars, lens, M)


@pytest.mark.parametrize("names", [CFG_1, CFG_3, CFG_A, CFG_23, CFG_123], ids=["r1", "r3", "r1r2", "r2r3", "r1r2r3"])
def test_reveal_mask_stress_and_errors(oracle, names):
    chars, lens = synth.reveal_stress(600, 700, seed=11)
    st, msk = _check_batch(oracle, names, chars, lens, 704)
    assert msk.any()
    chars, lens = synth.reveal_stress(100, 2000, seed=12)
    _check_batch(oracle, names, chars, lens, 2003)               # unaligned row count
    chars, lens = synth.ragged(257, 300, seed=3)
    rng = np.random.default_rng(1)
    for b in range(0, 257, 3):
        if lens[b]:
            chars[b, int(rng.integers(0, lens[b]))] = 200 + b % 50   # bytes the DFAs have no transition for
    lens[5] = 400                                                    # n > M
    st, _ = _check_batch(oracle, names, chars, lens, 304)
    codes = set(int(s) & 0xff for s in st)
    assert 1 in codes and 3 in codes


def test_flag_overlap_and_full_length_strings(oracle):
    chars, lens = synth.ragged(300, 200, seed=4)
    st, _ = _check_batch(oracle, [CFG_1[0], CFG_1[0]], chars, lens, 200)     # the same def twice: every flag overlaps
    assert any((int(s) & 0xff) == 2 for s in st)
    chars, lens = synth.regex1_planted(64, 128, seed=1, stride=128)           # n == M: the final state row does not exist
    _check_batch(oracle, CFG_1, chars, lens, 128)
    lens[:] = 0                                                               # n == 0
    _check_batch(oracle, CFG_A, chars, lens, 128)


HDR = [["header_from_lookup.txt", ["header_from_substr0.txt"]], ["header_to_lookup.txt", ["header_to_substr0.txt"]],
       ["header_subject_lookup.txt", ["header_subject_substr%d.txt" % k for k in range(3)]]]
CFG_D4 = CFG_123 + [HDR[0]]
CFG_D5 = [CFG_3[0], HDR[2], CFG_1[0], HDR[1], CFG_A[1]]
CFG_D7 = CFG_123 + [CFG_EX[0]] + HDR                    # def 3 = the partial example DFA: undefined transitions on almost any noise
CFG_D8 = CFG_123 + HDR + [CFG_1[0], CFG_A[1]]           # regex1 and regex2 a second time: both copies flag the same rows


@pytest.mark.parametrize("names", [CFG_D4, CFG_D5, CFG_D7, CFG_D8], ids=["D4", "D5", "D7", "D8"])
def test_more_than_three_regex_defs(oracle, names):
    """regex_defs is a Vec of any length (src/lib.rs:112): substr ids keep counting across the defs (lib.rs:827,842), the accept
    mask has a bit per def, the lowest def's undefined transition wins, two defs flagging one row is reported with its row."""
    M = 328
    cfg = _cfg(names, M)
    assert [cfg.substr_id_offset(d) for d in range(len(names))] == list(np.cumsum([1] + [len(s) for _, s in names[:-1]]))
    o = OracleDefs.from_files(oracle, names)
    chars, lens = synth.reveal_stress(400, 320, seed=31)
    h_c, h_l = synth.headers_planted(200, 320, seed=3, stride=320)
    chars, lens = np.concatenate([chars, h_c]), np.concatenate([lens, h_l])
    chars[7, 50] = 250                                           # a byte no DFA has a transition for
    lens[11] = 400                                               # n > M
    st, msk = _check_batch(oracle, names, chars, lens, M, cfg=cfg, o=o)
    codes = st & np.uint64(0xff)
    assert (codes == 1).any() and (codes == 3).any()
    if names is CFG_D7:
        assert ((st[codes == 1] >> np.uint64(8)) & np.uint64(0xff) == 3).sum() > 300      # the example DFA (def 3) dies on noise; def 0..2 do not
    elif names is CFG_D8:
        assert (codes == 2).sum() > 300 and (codes == 0).sum() > 10                       # overlaps wherever a copied def tags a row
    else:
        assert (codes == 0).sum() > 100 and msk.any()               # (regex3 and header_from both tag a `from:` line: those strings are overlaps)
        assert len(set(int(x) for x in st[codes == 0] >> np.uint64(8))) > 2              # different subsets of the defs accept


def test_planted_cfg2_sample(oracle):
    """BASELINE configs[1]'s workload (regex1+substr1, n = 1023, M = 1024), a slice of it"""
    chars, lens = synth.regex1_planted(2048, 1023, seed=0, stride=1024)
    st, msk = _check_batch(oracle, CFG_1, chars, lens, 1024)
    assert (st == np.uint64(0x100)).mean() > 0.9 and msk.any()


def test_host_threshold_is_settable_and_host_only_configs_refuse_device_batches():
    cfg = _cfg(CFG_1, 64)
    assert cfg.host_threshold() == hra.HRX_DEFAULT_HOST_THRESHOLD
    cfg.set_host_threshold(1234)
    assert cfg.host_threshold() == 1234
    assert hra.lib.hrx_ctx_device(cfg._ctx) == hra.HRX_DEVICE_NONE
    rc = hra.lib.hrx_witness_batch_device(cfg._ctx, 16, 16, 16, 1, 64, 16, 16, 16, None)
    assert rc == hra.HRX_ERR_HIP


def test_random_definitions(oracle):
    """random DFAs / substring definitions in the reference's text formats (the generator of the GPU parity tests), D = 1..3"""
    from test_parity_gpu import _random_defs
    for seed in range(12):
        rng = np.random.default_rng(2000 + seed)
        D = 1 + seed % 3
        defs_t = _random_defs(rng, D)
        M = int(rng.choice([5, 31, 64, 100, 129, 256, 321]))
        B = 48
        stride = (M + 40 + 15) // 16 * 16
        alpha = np.unique(np.concatenate([a for _, _, a in defs_t]))
        common = defs_t[0][2]
        for _, _, a in defs_t[1:]:
            common = np.intersect1d(common, a)
        pool = common if len(common) >= 2 and rng.random() < 0.7 else alpha
        chars = pool[rng.integers(0, len(pool), size=(B, stride))].astype(np.uint8)
        lens = rng.integers(0, M + 1, size=B).astype(np.uint32)
        lens[rng.random(B) < 0.05] = M + 3                                      # BadLength
        lens[0] = M
        for b in np.nonzero(rng.random(B) < 0.1)[0]:                            # a byte no def has a column for
            chars[b, int(rng.integers(0, stride))] = 0
        defs = [hra.RegexDefs(hra.AllstrRegexDef(a), [hra.SubstrRegexDef(t) for t in subs]) for a, subs, _ in defs_t]
        cfg = hra.RegexVerifyConfig.configure(M, defs, device=hra.HRX_DEVICE_NONE)
        _check_batch(oracle, None, chars, lens, M, cfg=cfg, o=OracleDefs(oracle, [(a, subs) for a, subs, _ in defs_t]))


@pytest.mark.parametrize("names", [CFG_1, CFG_A, CFG_123], ids=["D1", "D2", "D3"])
def test_records_decode_to_what_derive_states_ids_flags_return(oracle, names):
    """SURVEY §8 f3, the part that can be pinned without a Rust toolchain: the batch fill feeds lib.rs:339-773 from the compact records instead of the three
    derive_* calls (hrx_witness_of_string; bindings/rust/hrx.rs WitnessOf is that call).  The decode must give EXACTLY what derive_states / derive_substr_ids / derive_is_start_end return
    (lib.rs:804-888; the oracle's restatement of them) — every reference test string, planted and stress strings, n = 0 and n = M included; the one value
    the records cannot hold (states[d][M] and the last transition's end flag when n == M) is one the reference computes and never assigns (lib.rs:388-418, 501)."""
    M = 160
    D = len(names)
    cfg = _cfg(names, M)
    o = OracleDefs.from_files(oracle, names)
    texts = [c["input"].encode() if isinstance(c.get("input"), str) else None for c in reference_cases()]
    texts = [t for t in texts if t is not None and len(t) <= M]
    c2, l2 = synth.reveal_stress(60, M, seed=77)
    texts += [bytes(c2[b, :l2[b]]) for b in range(60)] + [b"", bytes(c2[0, :1])]
    full = bytes(synth.regex1_planted(1, M, seed=5, stride=M)[0][0, :M])
    texts.append(full)                                       # n == M
    chars = np.zeros((len(texts), M), np.uint8)
    lens = np.zeros(len(texts), np.uint32)
    for b, t in enumerate(texts):
        chars[b, :len(t)] = np.frombuffer(t, np.uint8)
        lens[b] = len(t)
    rec, msk, st = cfg.witness_batch_host(chars, lens)
    checked = 0
    for b, t in enumerate(texts):
        if int(st[b]) & 0xff:
            continue                                         # an undefined transition: the reference panics in derive_states (lib.rs:817); no records to decode
        n = len(t)
        want_states = o.derive_states(t)
        want_sids = o.derive_substr_ids(want_states)
        want_st, want_en = o.derive_is_start_end(want_states, want_sids)
        g_states, g_sids, g_st, g_en = hra.witness_of_string(rec[b], n)      # hrx_witness_of_string: the C export bindings/rust/hrx.rs WitnessOf::new calls
        if n == M:                                           # what no cell ever holds
            want_states = want_states.copy(); want_states[:, M] = 0
            want_en = want_en.copy(); want_en[:, M] = False
        assert np.array_equal(g_states, want_states) and np.array_equal(g_sids, want_sids), t
        assert np.array_equal(g_st, want_st) and np.array_equal(g_en, want_en), t
        checked += 1
    assert checked >= 40


@pytest.mark.parametrize("names", [CFG_1, CFG_A, CFG_123], ids=["D1", "D2", "D3"])
def test_batch_columns_are_what_match_substrs_assigns_per_circuit(oracle, names):
    """hrx_witness_columns_host (the batch form): every advice column of every circuit of a batch, column-major, equals what the single-string surface — the
    oracle's restatement of lib.rs:339-519, 593-764 — gives for that string; string-major and position-major host buffers, a sub-range of the batch, n = 0 and n = M."""
    M = 96
    D = len(names)
    cfg = _cfg(names, M)
    o = OracleDefs.from_files(oracle, names)
    c2, l2 = synth.reveal_stress(150, M, seed=5)
    c2[3, :] = synth.regex1_planted(1, M, seed=9, stride=M)[0][0, :M]; l2[3] = M      # n == M
    l2[4] = 0
    rec, msk, st = cfg.witness_batch_host(c2, l2)
    ok = (st & np.uint64(0xff)) == 0
    assert ok.sum() > 100
    B = len(l2)
    cols = hra.witness_columns_host(c2, l2, rec, msk, M, D)
    assert cols.shape == (4 + 4 * D, B, M)
    names_of = ["enable", "character"] + sum([["state", "substr_id", "start_enable", "end_enable"]] * D, []) + ["masked_char", "masked_substr_id"]
    for b in range(B):
        if not ok[b]:
            continue
        w = o.match_substrs(bytes(c2[b, :l2[b]]), M)
        for c, key in enumerate(names_of):
            want = w[key] if np.ndim(w[key]) == 1 else w[key][(c - 2) // 4]
            assert np.array_equal(cols[c, b], want), (b, key)
    # a sub-range, and the same rows out of position-major buffers (what the device writes, copied out as it is)
    sub = hra.witness_columns_host(c2, l2, rec, msk, M, D, b_begin=17, b_count=40)
    assert np.array_equal(sub, cols[:, 17:57])
    q4, q8 = (M + 3) // 4, (M + 7) // 8
    rec_pm = np.zeros((q4, D, B, 4), np.uint32); msk_pm = np.zeros((q8, B, 8), np.uint16)
    rec_pm[:] = rec.reshape(B, q4, 4, D).transpose(1, 3, 0, 2)
    msk_pm[:] = msk.reshape(B, q8, 8).transpose(1, 0, 2)
    stride = c2.shape[1]
    c_pm = np.ascontiguousarray(c2.reshape(B, stride // 16, 16).transpose(1, 0, 2))
    pmc = hra.witness_columns_host(c_pm, l2, rec_pm, msk_pm, M, D, position_major=True, chars_pm_stride=stride, B=B)
    assert np.array_equal(pmc, cols)
    with pytest.raises(hra.HrxError):
        hra.witness_columns_host(c2, l2, rec, msk, M, D, b_begin=140, b_count=20)


def test_a_failed_clone_leaves_the_source_config_usable(oracle):
    """RegexVerifyConfig.clone made its shallow copy BEFORE hrx_ctx_clone — a failing clone (no such device) left a copy that shared the source's context,
    and the copy's finalizer destroyed it.  The source must keep working, and destroying it afterwards must be a single free."""
    import gc
    cfg = _cfg(CFG_1, 64)
    with pytest.raises(hra.HrxError):
        cfg.clone(device=99)
    gc.collect()
    r = cfg.match_substrs(b"email was meant for @y.")
    assert bytes(r.masked_characters[21:22].astype(np.uint8)) == b"y"
    c2 = cfg.clone()                      # a host-only clone of a host-only context
    assert np.array_equal(c2.match_substrs(b"email was meant for @y.").masked_characters, r.masked_characters)
    del cfg, c2
    gc.collect()


def test_host_route_options_on_a_host_only_context(oracle):
    """hrx_ctx_set_option: a host-only context walks everything on the host whatever HRX_OPT_HOST_ROUTE says; HRX_OPT_HOST_THREADS bounds its threads; results do not depend on either."""
    M = 128
    cfg = _cfg(CFG_A, M)
    chars, lens = synth.reveal_stress(700, M - 8, seed=3)
    ref = cfg.witness_batch_host(chars, lens)
    for threads in (1, 3):
        cfg.set_option(hra.OPT_HOST_THREADS, threads)
        for route in (hra.HOST_ROUTE_AUTO, hra.HOST_ROUTE_DEVICE, hra.HOST_ROUTE_HOST):
            cfg.set_option(hra.OPT_HOST_ROUTE, route)
            got = cfg.witness_batch_host(chars, lens)
            assert all(np.array_equal(a, b) for a, b in zip(got, ref))
    with pytest.raises(hra.HrxError):
        cfg.set_option(hra.OPT_HOST_ROUTE, 7)
    assert cfg.get_option(hra.OPT_HOST_THREADS) == 3
    # the allocator's option: on by default, copied by a clone, 0 / 1 only
    assert cfg.get_option(hra.OPT_PLACE_DRY_LAUNCH) == 1
    cfg.set_option(hra.OPT_PLACE_DRY_LAUNCH, 0)
    assert cfg.clone().get_option(hra.OPT_PLACE_DRY_LAUNCH) == 0
    with pytest.raises(hra.HrxError):
        cfg.set_option(hra.OPT_PLACE_DRY_LAUNCH, 2)


def test_host_walk_threads_survive_a_fork(oracle):
    """The native walk's worker threads live in a process-wide pool; a forked child (Python multiprocessing) inherits the pool object without its threads and must start its own."""
    cfg = _cfg(CFG_1, 128)
    chars, lens = synth.reveal_stress(4000, 120, seed=1)
    a = cfg.witness_batch_host(chars, lens)        # (the parent's pool has run a job)
    pid = os.fork()
    if pid == 0:
        try:
            b = cfg.witness_batch_host(chars, lens)
            os._exit(0 if all(np.array_equal(x, y) for x, y in zip(a, b)) else 3)
        except BaseException:
            os._exit(4)
    _, st = os.waitpid(pid, 0)
    assert os.WIFEXITED(st) and os.WEXITSTATUS(st) == 0
