"""Parity tests proper: the HIP path, called through the C ABI, against the oracle (bit-exact) and against the
reference's own known answers.  Run with -m gpu on an MI355X."""
import os
import zlib

import numpy as np
import pytest

from oracle_lib import OracleDefs, DFA_DIR, reference_cases

pytestmark = pytest.mark.gpu

NO_HOST = 0x20000000     # kDbgNoHost (csrc/hrx_kernel.hpp): single strings and small host batches go to the device as well


def _flags(bits=0):
    return str(int(bits) | NO_HOST)


@pytest.fixture(autouse=True)
def _device_paths_only(monkeypatch):
    """These tests are about the kernels: the library's native small-batch host walk (tests/test_host_walk.py) stays out
    of the way.  HRX_DEBUG_FLAGS is read once per context, and every test makes its configs after this point."""
    monkeypatch.setenv("HRX_DEBUG_FLAGS", _flags())

CFG_1 = [["regex1_test_lookup.txt", ["substr1_test_lookup.txt"]]]
CFG_A = [["regex1_test_lookup.txt", ["substr1_test_lookup.txt"]], ["regex2_test_lookup.txt", ["substr2_test_lookup.txt"]]]
CFG_3 = [["regex3_test_lookup.txt", ["substr3_test_lookup.txt"]]]
CFG_23 = [["regex2_test_lookup.txt", ["substr2_test_lookup.txt"]], ["regex3_test_lookup.txt", ["substr3_test_lookup.txt"]]]
CFG_123 = CFG_A + CFG_3
CFG_EX = [["ex_allstr.txt", ["ex_substr_id1.txt"]]]


@pytest.fixture(scope="module")
def hra():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    import halo2_regex_amd as m
    return m


def _cfg(hra, names, M):
    defs = [hra.RegexDefs(hra.AllstrRegexDef.read_from_text(os.path.join(DFA_DIR, a)),
                          [hra.SubstrRegexDef.read_from_text(os.path.join(DFA_DIR, s)) for s in subs]) for a, subs in names]
    return hra.RegexVerifyConfig.configure(M, defs, device=0)


def _check_batch(hra, oracle, names, chars, lens, M):
    cfg = _cfg(hra, names, M)
    o = OracleDefs.from_files(oracle, names)
    orec, omsk, ost = o.witness_batch(chars, lens, M)
    grec, gmsk, gst = cfg.witness_batch_host(chars, lens)
    assert np.array_equal(ost, gst)
    ok = (ost & np.uint64(0xff)) == 0
    assert np.array_equal(orec[ok], grec[ok])
    assert np.array_equal(omsk[ok], gmsk[ok])
    return ost, omsk


# ---------------------------------------------------------------------------------------------
# the reference's own tests, through the reference-shaped single-string surface
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", reference_cases(), ids=[c["name"] for c in reference_cases()])
def test_reference_known_answers(hra, oracle, case):
    M = case["max_chars_size"]
    cfg = _cfg(hra, case["defs"], M)
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
    # every column against the oracle's restatement of match_substrs
    o = OracleDefs.from_files(oracle, case["defs"]).match_substrs(inp, M)
    assert np.array_equal(result.all_enable_flags, o["enable"])
    assert np.array_equal(result.all_characters, o["character"])
    assert np.array_equal(result.states, o["state"])
    assert np.array_equal(result.substr_ids, o["substr_id"])
    assert np.array_equal(result.start_enables, o["start_enable"])
    assert np.array_equal(result.end_enables, o["end_enable"])
    assert np.array_equal(result.masked_characters, o["masked_char"])
    assert np.array_equal(result.all_substr_ids, o["masked_substr_id"])


@pytest.mark.parametrize("case", reference_cases()[:6], ids=[c["name"] for c in reference_cases()[:6]])
def test_derive_functions_match_lib_rs_804_888(hra, oracle, case):
    cfg = _cfg(hra, case["defs"], case["max_chars_size"])
    o = OracleDefs.from_files(oracle, case["defs"])
    inp = case["input"].encode("latin-1")
    states = cfg.derive_states(inp)
    assert states.shape == (cfg.num_defs, len(inp) + 1) and np.array_equal(states, o.derive_states(inp))
    sids = cfg.derive_substr_ids(states)
    assert np.array_equal(sids, o.derive_substr_ids(states))
    st, en = cfg.derive_is_start_end(states, sids)
    ost, oen = o.derive_is_start_end(states, sids)
    assert np.array_equal(st, ost) and np.array_equal(en, oen)


def test_invalid_transition_panics_with_the_reference_message(hra):
    cfg = _cfg(hra, CFG_EX, 128)
    with pytest.raises(hra.HrxError, match=r"^The transition from 2 by 33 is invalid!$") as e:   # lib.rs:817
        cfg.derive_states(b"email was meant for @vitalik.!")
    assert e.value.code == hra.HRX_ERR_INVALID_TRANSITION
    with pytest.raises(hra.HrxError, match=r"The transition from 0 by 200 is invalid!"):
        _cfg(hra, CFG_1, 64).match_substrs(bytes([200]))
    assert _cfg(hra, CFG_1, 64).derive_states(b"").tolist() == [[0]]


def test_cpp_host_mirror_runs_the_reference_test(hra):
    """tests/host_cpp/test_host.cpp = test_substr_pass1 (lib.rs:1067-1092) through csrc/hrx_host.hpp"""
    import subprocess
    from oracle_lib import ROOT
    exe = "/tmp/hrx_test_host_gpu"
    csrc = os.path.join(ROOT, "halo2_regex_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "host_cpp", "test_host.cpp"), "-o", exe,
                           "-L" + csrc, "-lhrx", "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"])
    out = subprocess.run([exe, DFA_DIR, "gpu"], capture_output=True, text=True)
    assert out.returncode == 0 and "gpu ok" in out.stdout, out.stdout + out.stderr


# ---------------------------------------------------------------------------------------------
# batches against the oracle
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M", [1, 7, 8, 63, 64, 65, 72, 128, 200, 1024])
def test_ragged_batches_every_row_count(hra, oracle, M):
    from halo2_regex_amd import synth
    chars, lens = synth.ragged(200, M, seed=M)
    _check_batch(hra, oracle, CFG_1, chars, lens, M)
    _check_batch(hra, oracle, CFG_A, chars, lens, M)


@pytest.mark.parametrize("names", [CFG_1, CFG_3, CFG_A, CFG_23, CFG_123], ids=["r1", "r3", "r1r2", "r2r3", "r1r2r3"])
def test_reveal_mask_stress(hra, oracle, names):
    from halo2_regex_amd import synth
    chars, lens = synth.reveal_stress(1500, 700, seed=11)
    st, msk = _check_batch(hra, oracle, names, chars, lens, 704)
    assert msk.any()
    chars, lens = synth.reveal_stress(300, 2000, seed=12)
    _check_batch(hra, oracle, names, chars, lens, 2003)               # unaligned row count


def _check_batch_pm(hra, oracle, names, chars, lens, M):
    """the same batch through HRX_LAYOUT_POSITION_MAJOR (loader/walker kernel), de-permuted and compared bit for bit"""
    import torch
    cfg = _cfg(hra, names, M)
    o = OracleDefs.from_files(oracle, names)
    orec, omsk, ost = o.witness_batch(chars, lens, M)
    dev = torch.device("cuda", 0)
    B, D = len(lens), cfg.num_defs
    stride = (chars.shape[1] + 15) // 16 * 16
    wide = torch.zeros((B, max(stride, 16)), dtype=torch.uint8, device=dev)
    wide[:, :chars.shape[1]] = torch.from_numpy(chars).to(dev)
    d_lens = torch.from_numpy(lens.astype(np.int32)).to(dev)
    rec, msk, st = cfg.witness_batch_position_major(wide, d_lens)
    # and with the input chunked position-major as well: the same bytes must come out
    rec_i, msk_i, st_i = cfg.witness_batch_position_major(hra.chars_to_position_major(wide), d_lens, chars_pm_stride=wide.shape[1])
    torch.cuda.synchronize()
    okm = torch.from_numpy(((ost & np.uint64(0xff)) == 0)).to(dev)
    a_r, a_m = hra.position_major_to_string_major(rec, msk, B, M, D)
    b_r, b_m = hra.position_major_to_string_major(rec_i, msk_i, B, M, D)
    assert torch.equal(st, st_i) and torch.equal(a_r[okm], b_r[okm]) and torch.equal(a_m[okm], b_m[okm])
    grec, gmsk = hra.position_major_to_string_major(rec, msk, B, M, D)
    grec = grec.cpu().numpy().view(np.uint32)
    gmsk = gmsk.cpu().numpy().view(np.uint16)
    gst = st.cpu().numpy().view(np.uint64)
    assert np.array_equal(ost, gst)
    ok = (ost & np.uint64(0xff)) == 0
    assert np.array_equal(orec[ok], grec[ok])
    assert np.array_equal(omsk[ok], gmsk[ok])


@pytest.mark.parametrize("M", [1, 7, 8, 63, 64, 65, 72, 128, 200, 1024])
def test_position_major_ragged(hra, oracle, M):
    from halo2_regex_amd import synth
    chars, lens = synth.ragged(200, M, seed=M)
    _check_batch_pm(hra, oracle, CFG_1, chars, lens, M)
    _check_batch_pm(hra, oracle, CFG_A, chars, lens, M)


@pytest.mark.parametrize("names", [CFG_1, CFG_3, CFG_A, CFG_23, CFG_123], ids=["r1", "r3", "r1r2", "r2r3", "r1r2r3"])
def test_position_major_reveal_stress_and_errors(hra, oracle, names):
    from halo2_regex_amd import synth
    chars, lens = synth.reveal_stress(1500, 700, seed=11)
    _check_batch_pm(hra, oracle, names, chars, lens, 704)
    chars, lens = synth.reveal_stress(300, 2000, seed=12)
    _check_batch_pm(hra, oracle, names, chars, lens, 2003)
    chars, lens = synth.ragged(257, 300, seed=3)
    rng = np.random.default_rng(1)
    for b in range(0, 257, 3):
        if lens[b]:
            chars[b, int(rng.integers(0, lens[b]))] = 200 + b % 50
    lens[5] = 400
    _check_batch_pm(hra, oracle, names, chars, lens, 304)


def test_position_major_full_size_cfg2(hra, oracle):
    import torch
    from halo2_regex_amd import synth
    B, n, M = 65536, 1023, 1024
    chars, lens = synth.regex1_planted(B, n, seed=0, stride=1024)
    cfg = _cfg(hra, CFG_1, M)
    dev = torch.device("cuda", 0)
    d_chars, d_lens = torch.from_numpy(chars).to(dev), torch.from_numpy(lens.astype(np.int32)).to(dev)
    rec, msk, st = cfg.witness_batch_position_major(d_chars, d_lens)
    r0, m0, s0 = cfg.witness_batch(d_chars, d_lens)         # the string-major kernel on the same batch
    torch.cuda.synchronize()
    r1, m1 = hra.position_major_to_string_major(rec, msk, B, M, 1)
    assert torch.equal(r1, r0) and torch.equal(m1, m0) and torch.equal(st, s0)
    idx = np.random.default_rng(1).choice(B, 512, replace=False)
    orec, omsk, ost = OracleDefs.from_files(oracle, CFG_1).witness_batch(chars[idx], lens[idx], M)
    assert np.array_equal(orec, r1.cpu().numpy().view(np.uint32)[idx]) and np.array_equal(omsk, m1.cpu().numpy().view(np.uint16)[idx])
    rec2, msk2, st2 = cfg.witness_batch_position_major(d_chars, d_lens)
    torch.cuda.synchronize()
    assert torch.equal(rec, rec2) and torch.equal(msk, msk2) and torch.equal(st, st2)


def test_large_dfa_walks_out_of_global_memory(hra, oracle):
    """300- and 500-state DFAs (300/500 KiB of fused table) do not fit LDS: both layouts fall back to the global-table
    kernels; bit-exact against the oracle, including a partial DFA with undefined transitions."""
    from halo2_regex_amd import synth
    for nstates, total, seed in ((300, True, 2), (500, False, 3)):
        allstr, sub = synth.random_dfa(nstates, seed=seed, total=total)
        defs = [hra.RegexDefs(hra.AllstrRegexDef(allstr), [hra.SubstrRegexDef(sub)])]
        cfg = hra.RegexVerifyConfig.configure(264, defs, device=0)
        o = OracleDefs(oracle, [(allstr, [sub])])
        chars, lens = synth.ragged(300, 264, seed=nstates, planted=False)
        orec, omsk, ost = o.witness_batch(chars, lens, 264)
        grec, gmsk, gst = cfg.witness_batch_host(chars, lens)
        assert np.array_equal(ost, gst)
        ok = (ost & np.uint64(0xff)) == 0
        assert ok.any() and np.array_equal(orec[ok], grec[ok]) and np.array_equal(omsk[ok], gmsk[ok])
        if not total:
            assert (~ok).any()
        import torch
        dev = torch.device("cuda", 0)
        wide = torch.zeros((300, chars.shape[1]), dtype=torch.uint8, device=dev)
        wide[:] = torch.from_numpy(chars).to(dev)
        rec, msk, st = cfg.witness_batch_position_major(wide, torch.from_numpy(lens.astype(np.int32)).to(dev))
        torch.cuda.synchronize()
        r1, m1 = hra.position_major_to_string_major(rec, msk, 300, 264, 1)
        assert np.array_equal(st.cpu().numpy().view(np.uint64), ost)
        assert np.array_equal(r1.cpu().numpy().view(np.uint32)[ok], orec[ok]) and np.array_equal(m1.cpu().numpy().view(np.uint16)[ok], omsk[ok])


def _sample_check(hra, oracle_defs, cfg, chars, lens, M, D, nsample, seed, position_major=True):
    """device-resident run of the whole batch; a seeded sample of strings bit-exact against the oracle; idempotence"""
    import torch
    dev = torch.device("cuda", 0)
    B = len(lens)
    d_chars, d_lens = torch.from_numpy(chars).to(dev), torch.from_numpy(lens.astype(np.int32)).to(dev)
    if position_major:
        rec, msk, st = cfg.witness_batch_position_major(d_chars, d_lens)
        rec2, msk2, st2 = cfg.witness_batch_position_major(d_chars, d_lens)
        torch.cuda.synchronize()
        assert torch.equal(rec, rec2) and torch.equal(msk, msk2) and torch.equal(st, st2)
        rec, msk = hra.position_major_to_string_major(rec, msk, B, M, D)
    else:
        rec, msk, st = cfg.witness_batch(d_chars, d_lens)
        torch.cuda.synchronize()
    idx = np.sort(np.random.default_rng(seed).choice(B, nsample, replace=False))
    tidx = torch.from_numpy(idx).to(dev)
    orec, omsk, ost = oracle_defs.witness_batch(chars[idx], lens[idx], M)
    assert np.array_equal(ost, st[tidx].cpu().numpy().view(np.uint64))
    ok = (ost & np.uint64(0xff)) == 0
    assert np.array_equal(orec[ok], rec[tidx].cpu().numpy().view(np.uint32)[ok])
    assert np.array_equal(omsk[ok], msk[tidx].cpu().numpy().view(np.uint16)[ok])
    return st.cpu().numpy().view(np.uint64)


def test_cfg3_shape_two_defs_2048_byte_strings(hra, oracle):
    """BASELINE configs[2] shape (regex2 + regex3 with substr extraction, 2048-byte strings) at B = 32768, both layouts."""
    from halo2_regex_amd import synth
    B, n, M = 32768, 2047, 2048
    chars, lens = synth.regex23_planted(B, n, seed=1, stride=2048)
    lens[:64] = np.random.default_rng(0).integers(0, n + 1, 64)       # some ragged strings as well
    o = OracleDefs.from_files(oracle, CFG_23)
    cfg = _cfg(hra, CFG_23, M)
    st = _sample_check(hra, o, cfg, chars, lens, M, 2, 256, 3, position_major=True)
    assert (st & np.uint64(0xff) == 0).all()
    _sample_check(hra, o, cfg, chars, lens, M, 2, 256, 4, position_major=False)


def test_cfg5_shape_256_state_dense_dfa_4096_byte_strings(hra, oracle):
    """BASELINE configs[4] shape: synthetic total DFA, 256 states x 256 symbols (258 KiB of 4-byte fused table: the
    position-major kernel walks the 128-KiB HALF table in LDS, the string-major one the global table), 4096-byte inputs over
    all byte values."""
    from halo2_regex_amd import synth
    allstr, sub = synth.random_dfa(256, seed=2, alphabet=np.arange(256, dtype=np.uint8), n_substr_pairs=200)
    defs = [hra.RegexDefs(hra.AllstrRegexDef(allstr), [hra.SubstrRegexDef(sub)])]
    B, n, M = 4096, 4095, 4096
    cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
    assert cfg.table_bytes() == 258 * 1024
    chars, lens = synth.noise(B, n, seed=2, alphabet=np.arange(256, dtype=np.uint8), stride=4096)
    o = OracleDefs(oracle, [(allstr, [sub])])
    st = _sample_check(hra, o, cfg, chars, lens, M, 1, 96, 5, position_major=True)
    assert (st & np.uint64(0xff) == 0).all()
    _sample_check(hra, o, cfg, chars, lens, M, 1, 96, 6, position_major=False)


def test_half_table_kernel_on_dfas_of_up_to_256_states(hra, oracle):
    """DFAs of 141..256 states: the 4-byte fused table does not fit LDS next to the input rings, the 2-byte HALF table does
    (position-major kernel).  Total and partial DFAs (undefined transitions -> status 1 with the reference's state/char),
    ragged lengths, bytes outside the alphabet."""
    import torch
    from halo2_regex_amd import synth
    dev = torch.device("cuda", 0)
    for nstates, total, seed, alpha in ((200, False, 7, synth.ALPHABET98), (256, True, 2, np.arange(256, dtype=np.uint8)),
                                        (160, False, 9, np.arange(256, dtype=np.uint8))):
        allstr, sub = synth.random_dfa(nstates, seed=seed, total=total, alphabet=alpha, n_substr_pairs=120)
        defs = [hra.RegexDefs(hra.AllstrRegexDef(allstr), [hra.SubstrRegexDef(sub)])]
        M = 328
        cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
        assert cfg.table_bytes() == (nstates + 2) * 1024
        o = OracleDefs(oracle, [(allstr, [sub])])
        chars, lens = synth.ragged(700, M, seed=nstates, planted=False, alphabet=alpha)
        if len(alpha) < 256:
            chars[3, 17] = 200
            chars[4, 0] = 255
        orec, omsk, ost = o.witness_batch(chars, lens, M)
        ok = (ost & np.uint64(0xff)) == 0
        assert ok.any() and (total or (~ok).any())
        wide = torch.from_numpy(chars).to(dev)
        rec, msk, st = cfg.witness_batch_position_major(wide, torch.from_numpy(lens.astype(np.int32)).to(dev))
        torch.cuda.synchronize()
        r1, m1 = hra.position_major_to_string_major(rec, msk, 700, M, 1)
        assert np.array_equal(st.cpu().numpy().view(np.uint64), ost)
        assert np.array_equal(r1.cpu().numpy().view(np.uint32)[ok], orec[ok]) and np.array_equal(m1.cpu().numpy().view(np.uint16)[ok], omsk[ok])


def test_cfg4_shape_three_header_defs_five_substrs(hra, oracle):
    """BASELINE configs[3] shape: D=3 header-style definitions (from / to / subject; substr ids 1..5) compiled by this
    repo's own regex compiler; 2048-row strings in both layouts and 32768-byte strings, sampled against the oracle."""
    from halo2_regex_amd import synth
    from test_substr_gen import header_def_texts
    texts = header_def_texts()
    defs = [hra.RegexDefs(hra.AllstrRegexDef(a), [hra.SubstrRegexDef(t) for t in subs]) for a, subs in texts]
    o = OracleDefs(oracle, texts)
    for B, n, M, ns in ((4096, 2047, 2048, 128), (256, 32767, 32768, 12)):
        cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
        chars, lens = synth.headers_planted(B, n, seed=3, stride=M)
        st = _sample_check(hra, o, cfg, chars, lens, M, 3, ns, 11, position_major=True)
        assert ((st & np.uint64(0xff)) == 0).all() and (st == np.uint64(0x700)).mean() > 0.99    # a block planted at offset 0 is not accepted
        _sample_check(hra, o, cfg, chars, lens, M, 3, ns, 12, position_major=False)


def test_global_table_variant_on_the_reference_dfas(hra, oracle, monkeypatch):
    from halo2_regex_amd import synth
    monkeypatch.setenv("HRX_DEBUG_FLAGS", _flags(0x40000))
    chars, lens = synth.reveal_stress(500, 700, seed=31)
    _check_batch(hra, oracle, CFG_A, chars, lens, 704)
    _check_batch_pm(hra, oracle, CFG_123, chars, lens, 704)


@pytest.mark.parametrize("flags", [str(0x80000), str(0x200000 | 0x2000000), str(0x400000), str(0x4000000), "0", str(0x8000000), str(0x40000000)],
                         ids=["narrow-table", "wide-table", "half-table", "def-parallel", "planner-default", "no-pair-step", "pair-step"])
def test_position_major_kernel_on_both_table_formats(hra, oracle, flags, monkeypatch):
    """The position-major path has four table formats (4-byte, WIDE for D >= 2, HALF for big DFAs, PAIR — two bytes per lookup —
    for one def with few byte classes: the planner's default at D = 1) and, for D >= 2 batches that leave walker slots empty,
    a def-parallel kernel (one walker wave per def, flags combined through LDS); force each through the same batches, D = 1..3, including strings with undefined transitions, bytes >= 128 (no column in the WIDE
    table) and two defs flagging the same row."""
    from halo2_regex_amd import synth
    monkeypatch.setenv("HRX_DEBUG_FLAGS", _flags(int(flags, 0)))
    chars, lens = synth.reveal_stress(700, 700, seed=21)
    chars[5, 100] = 200                      # a byte the DFAs have no column for
    chars[6, 699] = 255
    chars[7, 650] = 128
    lens[7] = 600                            # ... beyond the string's length: must not matter
    for cfg in (CFG_1, CFG_23, CFG_123, CFG_A):
        _check_batch_pm(hra, oracle, cfg, chars, lens, 704)
    chars, lens = synth.ragged(300, 200, seed=4)
    _check_batch_pm(hra, oracle, CFG_123, chars, lens, 200)
    _check_batch_pm(hra, oracle, [CFG_1[0], CFG_1[0]], chars, lens, 200)     # the same def twice: every flag overlaps


@pytest.mark.parametrize("flags", ["65536", "196608"], ids=["one-wave-gs64", "one-wave-gs32"])
def test_one_wave_kernel_variants(hra, oracle, flags, monkeypatch):
    """D <= 2 normally takes the walker/storer kernel; force the one-wave kernel (used for D = 3 and unaligned M)
    through the same batches, at its regular and its small-batch group size."""
    from halo2_regex_amd import synth
    monkeypatch.setenv("HRX_DEBUG_FLAGS", _flags(int(flags, 0)))
    chars, lens = synth.reveal_stress(700, 700, seed=21)
    _check_batch(hra, oracle, CFG_1, chars, lens, 704)
    _check_batch(hra, oracle, CFG_23, chars, lens, 704)
    chars, lens = synth.ragged(300, 200, seed=4)
    _check_batch(hra, oracle, CFG_A, chars, lens, 200)


def _random_defs(rng, D, big=False):
    """D random definitions in the reference's text formats: 3..60 states, mostly over one shared alphabet and mostly total
    (so that walks are long), sometimes 85-95 % dense or over an alphabet of their own; 1-3 substring definitions each with
    random transition subsets / start / end states."""
    shared = np.sort(rng.choice(np.arange(1, 256), size=int(rng.integers(2, 24)), replace=False))
    out = []
    for _ in range(D):
        S = int(rng.integers(150, 320)) if big else int(rng.integers(3, 61))   # big: HALF table (<= 256 states) or global table
        alpha = shared if rng.random() < 0.8 else np.sort(rng.choice(np.arange(1, 256), size=int(rng.integers(2, 40)), replace=False))
        dens = float(rng.choice([1.0, 1.0, 1.0, 0.97, 0.9]))
        lines = [str(int(rng.integers(0, S))), str(int(rng.integers(0, S))), str(S - 1)]
        pairs = set()
        for st in range(S):
            for ch in alpha:
                if rng.random() < dens:
                    nx = int(rng.integers(0, S)) if rng.random() < 0.5 else int(rng.integers(0, min(S, 4)))   # a few hub states: pairs recur
                    lines.append("%d %d %d" % (st, nx, int(ch)))
                    pairs.add((st, nx))
        pairs = sorted(pairs)
        subs = []
        for _ in range(int(rng.integers(1, 4))):
            pick = [pairs[i] for i in rng.choice(len(pairs), size=min(len(pairs), int(rng.integers(4, 40))), replace=False)]
            starts = sorted({a for a, _ in pick[: max(1, len(pick) // 3)]})
            ends = sorted({b for _, b in pick[len(pick) // 2:]}) or [pick[0][1]]
            subs.append("\n".join(["8", "0", "99", " ".join(map(str, starts)) + " ", " ".join(map(str, ends)) + " "] +
                                  ["%d %d" % p for p in sorted(pick)]) + "\n")
        out.append(("\n".join(lines) + "\n", subs, alpha))
    return out


@pytest.mark.parametrize("seed", list(range(16)) + [100, 101, 102, 103, 104, 105])
def test_fuzz_random_definitions_shapes_and_layouts(hra, oracle, seed):
    """Seeded fuzz: random DFAs (partial ones included: status 1 must carry the reference's state/char), 1-3 defs (overlapping
    flags -> status 2), random M incl. odd values, ragged lengths incl. 0 and > M, bytes outside the alphabets; the
    string-major and the position-major kernels against the oracle, bit for bit."""
    import torch
    rng = np.random.default_rng(1000 + seed)
    big = seed >= 100                                                        # one big DFA: the HALF-table / global-table kernels
    D = 1 if big else int(rng.integers(1, 4))
    defs_t = _random_defs(rng, D, big)
    M = int(rng.choice([5, 31, 64, 100, 129, 256, 321, 520, 777]))
    B = int(rng.choice([1, 63, 64, 65, 200, 333, 500]))
    stride = (M + 40 + 15) // 16 * 16
    alpha = np.unique(np.concatenate([a for _, _, a in defs_t]))
    common = defs_t[0][2]
    for _, _, a in defs_t[1:]:
        common = np.intersect1d(common, a)
    pool = common if len(common) >= 2 and rng.random() < 0.7 else alpha     # mostly bytes every def knows: long valid walks
    chars = pool[rng.integers(0, len(pool), size=(B, stride))].astype(np.uint8)
    lens = rng.integers(0, M + 1, size=B).astype(np.uint32)
    lens[rng.random(B) < 0.05] = M + 3                                      # BadLength
    lens[0] = M if B > 1 else lens[0]
    for b in np.nonzero(rng.random(B) < 0.1)[0]:                            # a byte no def has a column for
        chars[b, int(rng.integers(0, stride))] = 0
    defs = [hra.RegexDefs(hra.AllstrRegexDef(a), [hra.SubstrRegexDef(t) for t in subs]) for a, subs, _ in defs_t]
    cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
    o = OracleDefs(oracle, [(a, subs) for a, subs, _ in defs_t])
    orec, omsk, ost = o.witness_batch(chars, lens, M)
    ok = (ost & np.uint64(0xff)) == 0
    grec, gmsk, gst = cfg.witness_batch_host(chars, lens)                   # string-major kernels
    assert np.array_equal(ost, gst)
    assert np.array_equal(orec[ok], grec[ok]) and np.array_equal(omsk[ok], gmsk[ok])
    dev = torch.device("cuda", 0)
    d_chars, d_lens = torch.from_numpy(chars).to(dev), torch.from_numpy(lens.astype(np.int32)).to(dev)
    for pm_input in (False, True):                                          # position-major kernel, both input layouts
        if pm_input:
            rec, msk, st = cfg.witness_batch_position_major(hra.chars_to_position_major(d_chars), d_lens, chars_pm_stride=stride)
        else:
            rec, msk, st = cfg.witness_batch_position_major(d_chars, d_lens)
        torch.cuda.synchronize()
        r1, m1 = hra.position_major_to_string_major(rec, msk, B, M, D)
        assert np.array_equal(st.cpu().numpy().view(np.uint64), ost)
        assert np.array_equal(r1.cpu().numpy().view(np.uint32)[ok], orec[ok]) and np.array_equal(m1.cpu().numpy().view(np.uint16)[ok], omsk[ok])


def test_three_defs_every_kernel_agrees_at_a_chip_filling_size(hra, oracle, monkeypatch):
    """D = 3, 16384 x 1024-byte strings (one group per CU): the def-parallel kernel (the planner's choice), the regular
    position-major kernel, its narrow-table build and the string-major path must produce the same bytes; a seeded sample
    against the oracle."""
    import torch
    from halo2_regex_amd import synth
    dev = torch.device("cuda", 0)
    B, n, M = 16384, 1023, 1024
    chars, lens = synth.regex23_planted(B, n, seed=5, stride=1024)
    lens[:128] = np.random.default_rng(2).integers(0, n + 1, 128)
    d_chars, d_lens = torch.from_numpy(chars).to(dev), torch.from_numpy(lens.astype(np.int32)).to(dev)
    results = {}
    for name, flags in (("def-parallel", "0"), ("regular", str(0x2000000)), ("narrow", str(0x2000000 | 0x80000))):
        monkeypatch.setenv("HRX_DEBUG_FLAGS", _flags(int(flags, 0)))
        cfg = _cfg(hra, CFG_123, M)
        kern = cfg.describe_launch(B, layout=1)
        assert ("pmd_kernel" in kern) == (name == "def-parallel")
        rec, msk, st = cfg.witness_batch_position_major(d_chars, d_lens)
        torch.cuda.synchronize()
        results[name] = hra.position_major_to_string_major(rec, msk, B, M, 3) + (st,)
    monkeypatch.setenv("HRX_DEBUG_FLAGS", _flags())
    results["string-major"] = _cfg(hra, CFG_123, M).witness_batch(d_chars, d_lens)
    torch.cuda.synchronize()
    ref = results["regular"]
    for name, r in results.items():
        assert torch.equal(r[0], ref[0]) and torch.equal(r[1], ref[1]) and torch.equal(r[2], ref[2]), name
    idx = np.sort(np.random.default_rng(3).choice(B, 256, replace=False))
    orec, omsk, ost = OracleDefs.from_files(oracle, CFG_123).witness_batch(chars[idx], lens[idx], M)
    tidx = torch.from_numpy(idx).to(dev)
    assert np.array_equal(ref[2][tidx].cpu().numpy().view(np.uint64), ost)
    assert np.array_equal(ref[0][tidx].cpu().numpy().view(np.uint32), orec) and np.array_equal(ref[1][tidx].cpu().numpy().view(np.uint16), omsk)


def test_position_major_buffers_are_blocked_by_65536_strings(hra, oracle):
    """B = 65536 + 4200: two blocks of the position-major buffers (a full one and a partial one), both input layouts, D = 1 and
    D = 2; the field-cell expansion reads the same blocked buffers."""
    import torch
    from halo2_regex_amd import synth
    dev = torch.device("cuda", 0)
    B, M = hra.PM_BLOCK + 4200, 72
    for names, D in ((CFG_1, 1), (CFG_A, 2)):
        chars, lens = synth.ragged(B, M, seed=17)
        cfg = _cfg(hra, names, M)
        o = OracleDefs.from_files(oracle, names)
        orec, omsk, ost = o.witness_batch(chars, lens, M)
        ok = (ost & np.uint64(0xff)) == 0
        d_chars, d_lens = torch.from_numpy(chars).to(dev), torch.from_numpy(lens.astype(np.int32)).to(dev)
        for pm_input in (False, True):
            src = hra.chars_to_position_major(d_chars) if pm_input else d_chars
            kw = dict(chars_pm_stride=chars.shape[1]) if pm_input else {}
            rec, msk, st = cfg.witness_batch_position_major(src, d_lens, **kw)
            torch.cuda.synchronize()
            r1, m1 = hra.position_major_to_string_major(rec, msk, B, M, D)
            assert np.array_equal(st.cpu().numpy().view(np.uint64), ost)
            assert np.array_equal(r1.cpu().numpy().view(np.uint32)[ok], orec[ok]) and np.array_equal(m1.cpu().numpy().view(np.uint16)[ok], omsk[ok])
            # the first string of the second block, by the documented formula
            b, r, d = hra.PM_BLOCK, 5, D - 1
            nb, q4 = B - hra.PM_BLOCK, (M + 3) // 4
            flat = rec.cpu().numpy().view(np.uint32)
            assert flat[hra.PM_BLOCK * q4 * D * 4 + ((r // 4 * D + d) * nb + (b - hra.PM_BLOCK)) * 4 + r % 4] == orec[b, r, d]
            cells = cfg.fr_columns(src, d_lens, (rec, msk, st), b_begin=hra.PM_BLOCK - 3, b_count=6, position_major=True, canonical=True, **kw)
            got = cells.cpu().numpy().view(np.uint64)[..., 0]
            assert np.array_equal(got[2], orec[hra.PM_BLOCK - 3:hra.PM_BLOCK + 3, :, 0] & 0xffff)                 # states[0]
            assert np.array_equal(got[2 + 4 * D], omsk[hra.PM_BLOCK - 3:hra.PM_BLOCK + 3] & 0xff)                  # masked_characters


def test_multi_device_driver_shards_by_string_index(hra, oracle):
    """hrx_multi_*: the batch is cut with hrx_shard_range and the shards run concurrently, one context each (here three
    contexts on the one device: the same code path as three devices); results equal the single-call ones, including the
    empty trailing shard of a tiny batch and the error / bad-length statuses."""
    from halo2_regex_amd import synth
    chars, lens = synth.reveal_stress(1000, 300, seed=9)
    chars[7, 50] = 250
    lens[11] = 400
    cfg = _cfg(hra, CFG_A, 304)
    one = cfg.witness_batch_host(chars, lens)
    multi = hra.MultiDevice(cfg, [0, 0, 0])
    assert multi.num_shards == 3
    many = multi.witness_batch_host(chars, lens)
    for x, y in zip(one, many):
        assert np.array_equal(x, y)
    orec, omsk, ost = OracleDefs.from_files(oracle, CFG_A).witness_batch(chars, lens, 304)
    assert np.array_equal(many[2], ost) and (ost & np.uint64(0xff) == 1).any() and (ost & np.uint64(0xff) == 3).any()
    two = multi.witness_batch_host(chars[:2], lens[:2])                  # 2 strings over 3 shards: the last shard is empty
    assert np.array_equal(two[0], one[0][:2]) and np.array_equal(two[1], one[1][:2]) and np.array_equal(two[2], one[2][:2])


def test_invalid_bytes_bad_lengths_and_overlap_status(hra, oracle):
    from halo2_regex_amd import synth
    chars, lens = synth.ragged(257, 300, seed=3)
    rng = np.random.default_rng(1)
    for b in range(0, 257, 3):
        if lens[b]:
            chars[b, int(rng.integers(0, lens[b]))] = 200 + b % 50
    lens[5] = 400
    st, _ = _check_batch(hra, oracle, CFG_A, chars, lens, 304)
    codes = st & np.uint64(0xff)
    assert (codes == 1).any() and (codes == 3).any() and (codes == 0).any()
    inp = b"email was meant for @ab."
    c = np.zeros((2, 32), np.uint8)
    c[0, :len(inp)] = np.frombuffer(inp, np.uint8)
    st, _ = _check_batch(hra, oracle, [CFG_1[0], CFG_1[0]], c, np.array([len(inp), 0], np.uint32), 64)
    assert int(st[0]) & 0xff == 2


def test_planted_configs_of_baseline(hra, oracle):
    from halo2_regex_amd import synth
    chars, lens = synth.regex1_planted(4096, 1023, seed=0, stride=1024)      # cfg 2(b) shape, small batch
    st, msk = _check_batch(hra, oracle, CFG_1, chars, lens, 1024)
    assert (msk != 0).any(axis=1).mean() > 0.9                               # almost every string reveals its plant
    chars, lens = synth.regex23_planted(2048, 2047, seed=1, stride=2048)     # cfg 3 shape, small batch
    _check_batch(hra, oracle, CFG_23, chars, lens, 2048)
    chars, lens = synth.noise(1024, 1024, seed=0)                            # n == M
    _check_batch(hra, oracle, CFG_1, chars, lens, 1024)


# ---------------------------------------------------------------------------------------------
# BASELINE cfg 2 at full size: device-resident path + size-independent properties
# ---------------------------------------------------------------------------------------------
def test_full_size_cfg2_device_resident(hra, oracle):
    import torch
    from halo2_regex_amd import synth
    B, n, M = 65536, 1023, 1024
    chars, lens = synth.regex1_planted(B, n, seed=0, stride=1024)
    cfg = _cfg(hra, CFG_1, M)
    dev = torch.device("cuda", 0)
    d_chars = torch.from_numpy(chars).to(dev)
    d_lens = torch.from_numpy(lens.astype(np.int32)).to(dev)
    rec, msk, st = cfg.witness_batch(d_chars, d_lens)
    torch.cuda.synchronize()
    rec_h = rec.cpu().numpy().view(np.uint32)
    msk_h = msk.cpu().numpy().view(np.uint16)
    st_h = st.cpu().numpy().view(np.uint64)
    assert (st_h & np.uint64(0xff) == 0).all()
    # (1) a seeded sample of strings, bit-exact against the oracle
    o = OracleDefs.from_files(oracle, CFG_1)
    idx = np.random.default_rng(0).choice(B, 768, replace=False)
    orec, omsk, ost = o.witness_batch(chars[idx], lens[idx], M)
    assert np.array_equal(orec, rec_h[idx]) and np.array_equal(omsk, msk_h[idx]) and np.array_equal(ost, st_h[idx])
    # (2) idempotence: a second launch into fresh buffers gives the same bytes (checksum of checksums)
    rec2, msk2, st2 = cfg.witness_batch(d_chars, d_lens)
    torch.cuda.synchronize()
    assert torch.equal(rec, rec2) and torch.equal(msk, msk2) and torch.equal(st, st2)
    # (3) strings are independent: any shard of the batch reproduces its slice of the whole (the multi-GPU rule)
    for world in (2, 8):
        for rank in (0, world - 1):
            b, c = hra.shard_range(B, world, rank)
            r3, m3, s3 = cfg.witness_batch(d_chars[b:b + c].contiguous(), d_lens[b:b + c].contiguous())
            torch.cuda.synchronize()
            assert torch.equal(r3, rec[b:b + c]) and torch.equal(m3, msk[b:b + c]) and torch.equal(s3, st[b:b + c])
    # (3b) pitched buffers (hrx_recommended_pitches: non-power-of-two strides) hold the same rows
    rp, mp, cs = hra.recommended_pitches(M)
    assert rp >= M and mp >= M and rp % 8 == 0 and mp % 64 == 0 and cs % 16 == 0 and cs >= M
    wide = torch.zeros((B, cs), dtype=torch.uint8, device=dev)
    wide[:, :1024] = d_chars
    outp = cfg.alloc_outputs(B, dev, pitched=True)
    assert outp[0].stride(0) == rp and outp[1].stride(0) == mp
    r4, m4, s4 = cfg.witness_batch(wide, d_lens, out=outp)
    torch.cuda.synchronize()
    assert torch.equal(r4, rec) and torch.equal(m4, msk) and torch.equal(s4, st)
    # (4) structural properties of every row of every string (App. A.2): padding rows, masked only where tagged
    state = rec_h[:, :, 0] & 0xffff
    sid = (rec_h[:, :, 0] >> 16) & 0xff
    assert (state[:, n + 1:] == 29).all() and (sid[:, n:] == 0).all()
    assert (state[:, 0] == 0).all()
    assert ((msk_h != 0) <= (sid != 0)).all()
    assert ((msk_h >> 8)[msk_h != 0] == 1).all()
    assert ((msk_h & 0xff)[msk_h != 0] == chars[:, :M][msk_h != 0]).all()
