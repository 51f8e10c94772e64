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


@pytest.mark.parametrize("B,stride", [(1, 16), (63, 48), (64, 1024), (65, 272), (200, 4096), (70000, 64)])
def test_chars_to_position_major_device_equals_the_layout_definition(hra, B, stride):
    """hrx_chars_to_position_major_device (the library's route from the reference's input shape, one contiguous string per row, lib.rs:311-315, to
    HRX_LAYOUT_INPUT_POSITION_MAJOR) against the layout's definition (blocks of 65536 strings, [stride/16][nb][16]) — and a witness launch on its output
    against the launch on the string-major bytes."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(B * 131 + stride)
    chars = torch.from_numpy(rng.integers(0, 256, size=(B, stride), dtype=np.uint8)).to(dev)
    cfg = _cfg(hra, CFG_1, max(16, min(stride, 64)))
    got = cfg.chars_to_position_major_device(chars)
    torch.cuda.synchronize()
    want = hra.chars_to_position_major(chars)
    assert got.shape == want.shape and torch.equal(got, want)
    lens = torch.from_numpy(rng.integers(0, cfg.max_chars_size + 1, size=B).astype(np.int32)).to(dev)
    a = cfg.witness_batch_position_major(chars, lens)
    b = cfg.witness_batch_position_major(got, lens, chars_pm_stride=stride)
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_c_struct_entry_points_on_the_device(hra):
    """tests/host_c/test_push_structs.c (the Rust binding's call sequence: shuffled hrx_defs_push_allstr entries with explicit line indices, a duplicate key,
    hrx_defs_push_substr) with the witness rows of both configs computed on the device"""
    import subprocess
    from oracle_lib import ROOT
    exe = "/tmp/hrx_test_push_structs_gpu"
    csrc = os.path.join(ROOT, "halo2_regex_amd", "csrc")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "host_c", "test_push_structs.c"), "-o", exe,
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


def test_position_major_full_size_cfg2(hra, oracle, monkeypatch):
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
    del r0, m0, s0, r1, m1, rec, msk, st
    # every string of the batch against the oracle and through the integer MockProver — the planner's kernel for this shape
    # (one byte per lookup) and the pair-step kernel forced onto the same batch
    o = OracleDefs.from_files(oracle, CFG_1)
    for flags in (0, 0x40000000):
        monkeypatch.setenv("HRX_DEBUG_FLAGS", _flags(flags))
        cfg = _cfg(hra, CFG_1, M)
        assert ("witness_pp_kernel" in cfg.describe_launch(B, layout=3)) == (flags != 0)
        st = _full_check(hra, o, cfg, [(chars, lens)], M, 1, need_accept=0.9)
        assert (st & np.uint64(0xff) == 0).all()


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


def _pm_rows(hra, rec, msk, B, M, D, b0, b1):
    """strings [b0, b1) — inside one block of the position-major buffers — as string-major VIEWS (b1 - b0, M, D) / (b1 - b0, M)"""
    q4, q8 = (M + 3) // 4, (M + 7) // 8
    k0 = b0 // hra.PM_BLOCK * hra.PM_BLOCK
    nb = min(hra.PM_BLOCK, B - k0)
    assert b1 <= k0 + nb
    r = rec[k0 * q4 * 4 * D:(k0 + nb) * q4 * 4 * D].view(q4, D, nb, 4)[:, :, b0 - k0:b1 - k0]
    m = msk[k0 * q8 * 8:(k0 + nb) * q8 * 8].view(q8, nb, 8)[:, b0 - k0:b1 - k0]
    return r.permute(2, 0, 3, 1).reshape(b1 - b0, q4 * 4, D)[:, :M], m.permute(1, 0, 2).reshape(b1 - b0, q8 * 8)[:, :M]


def _full_check(hra, oracle_defs, cfg, blocks, M, D, position_major=True, elems=1 << 25, need_accept=None):
    """EVERY string of a device-resident batch against the oracle (all host cores), and through the integer MockProver
    (tests/mock_prover.py: the reference's gates, lookups and accept chain, src/lib.rs:126-305, 427-457).

    blocks: list of (chars, lens) host arrays, one per block of <= 65536 strings (the position-major buffers are blocked by
    that many).  The whole batch runs in ONE launch; the comparison then goes chunk by chunk (at most `elems` rows at a
    time): the oracle's rows for the chunk are uploaded and compared on the device with the kernel's rows, where they lie.
    Returns the status words."""
    import torch
    from mock_prover import IntegerMockProver, FAIL_ACCEPT
    dev = torch.device("cuda", 0)
    ncpu = os.cpu_count() or 1
    B = sum(len(l) for _, l in blocks)
    stride = blocks[0][0].shape[1]
    d_chars = torch.empty((B, stride), dtype=torch.uint8, device=dev)
    d_lens = torch.empty((B,), dtype=torch.int32, device=dev)
    k0 = 0
    for c, l in blocks:
        assert len(l) <= hra.PM_BLOCK and (len(l) == hra.PM_BLOCK or c is blocks[-1][0])
        d_chars[k0:k0 + len(l)] = torch.from_numpy(c).to(dev)
        d_lens[k0:k0 + len(l)] = torch.from_numpy(l.astype(np.int32)).to(dev)
        k0 += len(l)
    if position_major:
        src = hra.chars_to_position_major(d_chars)
        rec, msk, st = cfg.witness_batch_position_major(src, d_lens, chars_pm_stride=stride)
        rec2, msk2, st2 = cfg.witness_batch_position_major(src, d_lens, chars_pm_stride=stride)
        torch.cuda.synchronize()
        assert torch.equal(rec, rec2) and torch.equal(msk, msk2) and torch.equal(st, st2)      # idempotence
        del rec2, msk2, st2, src
    else:
        rec, msk, st = cfg.witness_batch(d_chars, d_lens)
        torch.cuda.synchronize()
    mp = IntegerMockProver.from_config(cfg, device=dev)
    all_accept = (1 << D) - 1
    chunk = max(64, min(hra.PM_BLOCK, elems // M))
    k0, n_ok, n_clean = 0, 0, 0
    for c, l in blocks:
        for c0 in range(0, len(l), chunk):
            c1 = min(len(l), c0 + chunk)
            orec, omsk, ost = oracle_defs.witness_batch(c[c0:c1], l[c0:c1], M, threads=ncpu)
            t_rec = torch.from_numpy(orec.view(np.int32)).to(dev)
            t_msk = torch.from_numpy(omsk.view(np.int16)).to(dev)
            t_st = torch.from_numpy(ost.view(np.int64)).to(dev)
            if position_major:
                g_rec, g_msk = _pm_rows(hra, rec, msk, B, M, D, k0 + c0, k0 + c1)
            else:
                g_rec, g_msk = rec[k0 + c0:k0 + c1], msk[k0 + c0:k0 + c1]
            g_st = st[k0 + c0:k0 + c1]
            assert torch.equal(g_st, t_st), "status words differ in strings %d..%d" % (k0 + c0, k0 + c1)
            ok = (t_st & 0xff) == 0                       # (status != 0: the reference panics / out of contract, rows unspecified)
            assert bool(((g_rec == t_rec).flatten(1).all(dim=1) | ~ok).all()), "records differ in strings %d..%d" % (k0 + c0, k0 + c1)
            assert bool(((g_msk == t_msk).all(dim=1) | ~ok).all()), "masked rows differ in strings %d..%d" % (k0 + c0, k0 + c1)
            # the constraint system itself, on the kernel's rows: every gate and lookup holds; where a def does not end in
            # its accept state the accept chain — and nothing else — fails (what MockProver::verify() reports for test_substr_fail*)
            code = mp.verify(d_chars[k0 + c0:k0 + c1], d_lens[k0 + c0:k0 + c1], g_rec, g_msk, M)
            accepted = (((t_st >> 8) & 0xff) == all_accept) | (d_lens[k0 + c0:k0 + c1] >= M)
            want = torch.where(accepted, torch.zeros_like(code), torch.full_like(code, FAIL_ACCEPT))
            bad = ok & (code != want)
            assert not bool(bad.any()), "MockProver: string %d: %s" % (k0 + c0 + int(bad.nonzero()[0]), mp.explain(code[bad][0]))
            n_ok += int(ok.sum())
            n_clean += int((ok & (code == 0)).sum())
            del t_rec, t_msk, t_st, code
        k0 += len(l)
    if need_accept is not None:
        assert n_clean >= need_accept * B, "only %d of %d strings satisfy the whole constraint system" % (n_clean, B)
    return st.cpu().numpy().view(np.uint64)


def test_full_check_detects_a_wrong_witness(hra, oracle):
    """negative control of the checker itself: the kernels' rows for regex1 compared with the oracle's rows for regex3 must fail"""
    from halo2_regex_amd import synth
    chars, lens = synth.regex1_planted(3000, 255, seed=2, stride=256)
    with pytest.raises(AssertionError):
        _full_check(hra, OracleDefs.from_files(oracle, CFG_3), _cfg(hra, CFG_1, 256), [(chars, lens)], 256, 1)


def _rolled_blocks(base_chars, base_lens, nblocks, last=None, seed=0):
    """nblocks blocks made from one generated block: block k = the base strings rotated by 977 k places, with 64 strings per
    block cut to a random shorter length (ragged rows in every block); `last`: strings in the last block (partial block)."""
    rng = np.random.default_rng(seed)
    out = []
    for k in range(nblocks):
        c = np.roll(base_chars, 977 * k, axis=0) if k else base_chars
        l = (np.roll(base_lens, 977 * k) if k else base_lens).copy()
        idx = rng.choice(len(l), min(64, len(l)), replace=False)
        l[idx] = rng.integers(0, l[idx] + 1)
        if last is not None and k == nblocks - 1:
            c, l = c[:last], l[:last]
        out.append((np.ascontiguousarray(c), l))
    return out


def test_cfg3_shape_two_defs_2048_byte_strings(hra, oracle):
    """BASELINE configs[2] shape (regex2 + regex3 with substr extraction, 2048-byte strings) at B = 32768, both layouts."""
    from halo2_regex_amd import synth
    B, n, M = 32768, 2047, 2048
    chars, lens = synth.regex23_planted(B, n, seed=1, stride=2048)
    lens[:64] = np.random.default_rng(0).integers(0, n + 1, 64)       # some ragged strings as well
    o = OracleDefs.from_files(oracle, CFG_23)
    cfg = _cfg(hra, CFG_23, M)
    st = _full_check(hra, o, cfg, [(chars, lens)], M, 2, position_major=True)
    assert (st & np.uint64(0xff) == 0).all()
    _full_check(hra, o, cfg, [(chars, lens)], M, 2, position_major=False)


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
    st = _full_check(hra, o, cfg, [(chars, lens)], M, 1, position_major=True)
    assert (st & np.uint64(0xff) == 0).all()
    _full_check(hra, o, cfg, [(chars, lens)], M, 1, position_major=False)


@pytest.mark.parametrize("flags", [0, 0x8000], ids=["byte-table", "half-table"])
def test_half_table_kernel_on_dfas_of_up_to_256_states(hra, oracle, flags, monkeypatch):
    """DFAs of 141..256 states: the 4-byte fused table does not fit LDS next to the input rings.  One def takes the BYTE table
    (1-byte next states on the chain, the pair tags off it: walker + loader + finisher), kDbgNoByte the 2-byte HALF table
    (walker + loader).  Total and partial DFAs (undefined transitions -> status 1 with the reference's state/char; 255 partial
    states fill the BYTE table's 256 rows), ragged lengths, bytes outside the alphabet."""
    import torch
    from halo2_regex_amd import synth
    dev = torch.device("cuda", 0)
    monkeypatch.setenv("HRX_DEBUG_FLAGS", _flags(flags))
    for nstates, total, seed, alpha in ((200, False, 7, synth.ALPHABET98), (256, True, 2, np.arange(256, dtype=np.uint8)),
                                        (160, False, 9, np.arange(256, dtype=np.uint8)), (255, False, 11, np.arange(256, dtype=np.uint8))):
        allstr, sub = synth.random_dfa(nstates, seed=seed, total=total, alphabet=alpha, n_substr_pairs=120)
        defs = [hra.RegexDefs(hra.AllstrRegexDef(allstr), [hra.SubstrRegexDef(sub)])]
        M = 328
        cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
        assert cfg.table_bytes() == (nstates + 2) * 1024
        assert ("witness_pm_kernel<1, false, false, true, false, false>" if flags else "witness_pm_kernel<1, false, false, false, false, true>") in cfg.describe_launch(700, layout=1)
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
    repo's own regex compiler; 2048-row strings and 32768-byte strings in both layouts, every string against the oracle."""
    from halo2_regex_amd import synth
    from test_substr_gen import header_def_texts
    texts = header_def_texts()
    defs = [hra.RegexDefs(hra.AllstrRegexDef(a), [hra.SubstrRegexDef(t) for t in subs]) for a, subs in texts]
    o = OracleDefs(oracle, texts)
    for B, n, M in ((4096, 2047, 2048), (256, 32767, 32768)):
        cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
        chars, lens = synth.headers_planted(B, n, seed=3, stride=M)
        st = _full_check(hra, o, cfg, [(chars, lens)], M, 3, position_major=True, need_accept=0.99)
        assert ((st & np.uint64(0xff)) == 0).all() and (st == np.uint64(0x700)).mean() > 0.99    # a block planted at offset 0 is not accepted
        _full_check(hra, o, cfg, [(chars, lens)], M, 3, position_major=False)


def test_global_table_variant_on_the_reference_dfas(hra, oracle, monkeypatch):
    from halo2_regex_amd import synth
    monkeypatch.setenv("HRX_DEBUG_FLAGS", _flags(0x40000))
    chars, lens = synth.reveal_stress(500, 700, seed=31)
    _check_batch(hra, oracle, CFG_A, chars, lens, 704)
    _check_batch_pm(hra, oracle, CFG_123, chars, lens, 704)


@pytest.mark.parametrize("flags", [str(0x80000), str(0x200000 | 0x2000000), str(0x400000), str(0x4000000), "0", str(0x8000000), str(0x40000000),
                               str(0x2000 | 0x8000000)],
                         ids=["narrow-table", "wide-table", "half-table", "def-parallel", "planner-default", "no-pair-step", "pair-step", "byte-table"])
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


@pytest.mark.parametrize("M", [512, 1024, 2048])
def test_chunked_launch_every_chunk_border_case(hra, oracle, M, monkeypatch):
    """The chunked launch (hrx_kernel_spec.hip: scout + compose find every chunk's start state, the loader / walker / finisher kernel
    walks the chunks as groups of their own, the stitch launch settles the reveal-mask carries and merges the status words), forced
    with chunks of 4 tiles (kDbgForceSpec) so that every string has 2 .. 8 of them: D = 1 .. 3, strings that end inside any chunk,
    at a chunk border, at M and at 0; the absorbing accept state behind a finished match (no warm-up from the start state reaches
    it); revealed substrings that straddle a chunk border (the start-mask assumption of the chunk behind it is wrong: its masked rows
    are recomputed); undefined transitions in a later chunk with the reference's (state, char) (lib.rs:817); two defs flagging one row."""
    from halo2_regex_amd import synth
    monkeypatch.setenv("HRX_DEBUG_FLAGS", _flags(0x80))
    chars, lens = synth.reveal_stress(700, M, seed=40 + M)
    rng = np.random.default_rng(M)
    for b in range(0, 700, 7):                       # full-length strings, some ending exactly at a chunk border
        lens[b] = M if b % 14 else 256 * (1 + b % 2)
        tail = synth.reveal_stress(1, M, seed=b)[0][0]
        chars[b, :M] = np.where(chars[b, :M] == 0, tail[:M], chars[b, :M])
    for b in range(3, 700, 31):                      # a byte no DFA has a column for, beyond the first chunk where the string is long enough
        if lens[b] > 300:
            chars[b, int(rng.integers(257, lens[b]))] = 200 + b % 50
    for b in range(5, 700, 53):                      # a revealed part that certainly crosses the row-256 border
        t = b"email was meant for @" + bytes(synth.LOWER[rng.integers(0, 26, size=60)]) + b"."
        chars[b, 220:220 + len(t)] = np.frombuffer(t, np.uint8)
        lens[b] = max(lens[b], 220 + len(t))
    for names in (CFG_1, CFG_23, CFG_123, CFG_A, [CFG_1[0], CFG_1[0]]):
        assert "chunked=%dx4 tiles" % (M // 256) in _cfg(hra, names, M).describe_launch(700, layout=3)
        _check_batch_pm(hra, oracle, names, chars, lens, M)


def test_chunked_launch_is_what_a_batch_of_few_long_strings_gets(hra, oracle):
    """Up to one group of 64 strings per CU and 4096 rows or more: the planner cuts the strings into chunks of 16 tiles (a launch of the
    sequential kernels would last one string's dependent chain, n x 29-50 ns, with three quarters of the walker slots empty).
    4096 x 8192-byte strings = 8 chunks each, every string against the oracle and the MockProver; D = 1 and D = 3."""
    from halo2_regex_amd import synth
    M = 8192
    for names, D, gen, seed in ((CFG_1, 1, synth.regex1_planted, 3), (CFG_123, 3, synth.noise, 4)):
        cfg = _cfg(hra, names, M)
        assert "chunked=8x16 tiles" in cfg.describe_launch(4096, layout=3) and "chunked" not in cfg.describe_launch(32768, layout=3)
        chars, lens = gen(4096, M - 1, seed=seed, stride=M)
        lens[::5] = np.random.default_rng(seed).integers(0, M, size=len(lens[::5]))          # ragged
        st = _full_check(hra, OracleDefs.from_files(oracle, names), cfg, [(chars, lens)], M, D)
        assert len(st) == 4096


@pytest.mark.parametrize("M,want", [(65536, "chunked=32x32 tiles"), (131072, "chunked=32x64 tiles")])
def test_chunked_launch_with_longer_chunks(hra, oracle, M, want):
    """Beyond 32768 rows the chunks grow (32 chunks of 32 / 64 tiles: the stitch launch holds 32 summaries per string, the repair wave a
    chunk of up to 64 tiles in LDS): 70 strings, ragged, planted matches, every string against the oracle and the MockProver."""
    from halo2_regex_amd import synth
    cfg = _cfg(hra, CFG_1, M)
    assert want in cfg.describe_launch(70, layout=3)
    chars, lens = synth.regex1_planted(70, M - 1, seed=M, stride=M)
    lens[::3] = np.random.default_rng(M).integers(0, M, size=len(lens[::3]))
    lens[1] = M
    st = _full_check(hra, OracleDefs.from_files(oracle, CFG_1), cfg, [(chars, lens)], M, 1)
    assert len(st) == 70


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


def _fuzz_seeds():
    """the suite's seeds, plus ranges named in HRX_FUZZ_EXTRA ("16:400,106:200") for a soak run (profiles/r02_soak.txt)"""
    seeds = list(range(16)) + [100, 101, 102, 103, 104, 105] + [20000, 20001, 20002, 20003, 20004, 20005] + [30000, 30001, 30002, 30003, 30004, 30005]
    for part in os.environ.get("HRX_FUZZ_EXTRA", "").split(","):
        if ":" in part:
            a, b = part.split(":")
            seeds += [x for x in range(int(a), int(b)) if x not in seeds]
    return seeds


@pytest.mark.parametrize("seed", _fuzz_seeds())
def test_fuzz_random_definitions_shapes_and_layouts(hra, oracle, seed):
    """Seeded fuzz: random DFAs (partial ones included: status 1 must carry the reference's state/char), 1-3 defs — 4-7 for the
    seeds from 20000 on: the multi-pass path — (overlapping flags -> status 2), random M incl. odd values, ragged lengths incl. 0 and > M, bytes outside the alphabets; the
    string-major and the position-major kernels against the oracle, bit for bit."""
    import torch
    rng = np.random.default_rng(1000 + seed)
    big = 100 <= seed < 10000                                                # one big DFA: the HALF-table / global-table kernels (soak seeds >= 10000: small again)
    D = 1 if big else (int(rng.integers(4, 8)) if seed >= 20000 else int(rng.integers(1, 4)))     # seeds >= 20000: more defs than one launch walks (passes + combine)
    if seed >= 30000:                                                        # ... up to HRX_MAX_DEFS = 32: more than four groups (always the combine launch), accept bits 8..39
        D = int(rng.choice([9, 13, 16, 21, 32]))
    defs_t = _random_defs(rng, D, big)
    if seed >= 30000:       # masked_substr_id is a u8: the ids of all defs' LAST substrings must sum to <= 255 (hrx_defs_finalize) — defs beyond that keep their DFA only
        off, total = 1, 0
        for k, (a_t, subs, al) in enumerate(defs_t):
            if total + off + len(subs) - 1 > 255 or rng.random() < 0.4:
                defs_t[k] = (a_t, [], al)
            else:
                total += off + len(subs) - 1
                off += len(subs)
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


def _spec_fuzz_seeds():
    """the suite's seeds, plus a range named in HRX_FUZZ_SPEC_EXTRA ("1000:1400") for a soak run (profiles/r03_soak.txt)"""
    seeds = list(range(300, 316)) + [100000, 100001, 100002, 100003] + [2499]     # 2499: a partial DFA whose quasi-absorbing survivor outlives a walked one (the scout once declared both dead)
    part = os.environ.get("HRX_FUZZ_SPEC_EXTRA", "")
    if ":" in part:
        a, b = part.split(":")
        seeds += [x for x in range(int(a), int(b)) if x not in seeds]
    return seeds


@pytest.mark.parametrize("seed", _spec_fuzz_seeds())
def test_fuzz_chunked_launch_on_random_definitions(hra, oracle, seed, monkeypatch):
    """Seeded fuzz of the chunked launch (forced, chunks of 4 tiles; seeds from 100000 on: the planner's own 16-tile chunks of 4096-8192 rows): random DFAs — partial ones included, and random transition
    functions need not forget their start state: chunks whose start states do not merge into the scout's bounds are walked by the
    compose launch — 1-3 defs, 2-8 chunks per string, ragged lengths incl. 0, M and > M, bytes outside the alphabets, flag overlap;
    against the oracle, bit for bit."""
    import torch
    natural = seed >= 100000                                  # the planner's own chunking (16 tiles) instead of forced 4-tile chunks
    monkeypatch.setenv("HRX_DEBUG_FLAGS", _flags(0 if natural else 0x80))
    rng = np.random.default_rng(5000 + seed)
    D = int(rng.integers(1, 4))
    defs_t = _random_defs(rng, D, False)
    M = int(rng.choice([4096, 5120, 8192])) if natural else int(rng.choice([512, 768, 1024, 1280, 2048]))
    B = int(rng.choice([1, 64, 65, 130])) if natural else int(rng.choice([1, 64, 65, 200, 333]))
    stride = M
    common = defs_t[0][2]
    for _, _, a in defs_t[1:]:
        common = np.intersect1d(common, a)
    alpha = np.unique(np.concatenate([a for _, _, a in defs_t]))
    pool = common if len(common) >= 2 else alpha
    chars = pool[rng.integers(0, len(pool), size=(B, stride))].astype(np.uint8)
    lens = rng.integers(0, M + 1, size=B).astype(np.uint32)
    lens[rng.random(B) < 0.3] = M
    lens[rng.random(B) < 0.1] = 256 * int(rng.integers(1, M // 256 + 1))   # ends exactly at a chunk border
    lens[rng.random(B) < 0.03] = M + 3                                      # BadLength
    for b in np.nonzero(rng.random(B) < 0.1)[0]:                            # a byte no def has a column for
        chars[b, int(rng.integers(0, stride))] = 0
    defs = [hra.RegexDefs(hra.AllstrRegexDef(a), [hra.SubstrRegexDef(t) for t in subs]) for a, subs, _ in defs_t]
    cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
    pm_input = seed % 3 != 1                                  # a third of the seeds hand the strings over string-major ([B][stride])
    assert "chunked=" in cfg.describe_launch(B, layout=3 if pm_input else 1) or cfg.table_bytes() > 100 * 1024     # (tables that leave no LDS for the rings are walked out of global memory, unchunked)
    orec, omsk, ost = OracleDefs(oracle, [(a, subs) for a, subs, _ in defs_t]).witness_batch(chars, lens, M)
    ok = (ost & np.uint64(0xff)) == 0
    dev = torch.device("cuda", 0)
    d_chars, d_lens = torch.from_numpy(chars).to(dev), torch.from_numpy(lens.astype(np.int32)).to(dev)
    if pm_input:
        rec, msk, st = cfg.witness_batch_position_major(hra.chars_to_position_major(d_chars), d_lens, chars_pm_stride=stride)
    else:
        rec, msk, st = cfg.witness_batch_position_major(d_chars, d_lens)
    torch.cuda.synchronize()
    r1, m1 = hra.position_major_to_string_major(rec, msk, B, M, D)
    assert np.array_equal(st.cpu().numpy().view(np.uint64), ost)
    assert np.array_equal(r1.cpu().numpy().view(np.uint32)[ok], orec[ok]) and np.array_equal(m1.cpu().numpy().view(np.uint16)[ok], omsk[ok])


@pytest.mark.parametrize("S", [7, 40])
def test_chunked_launch_of_a_dfa_that_never_forgets(hra, oracle, S):
    """A counter modulo S: every byte permutes the states, so no two start states ever merge and the scout gives up on EVERY
    chunk past the first (S > its survivor bound) — the compose launch's own walk (a 64th of the chunk per lane, from every
    state) carries the whole string.  The planner's natural chunk size (16 tiles), ragged lengths, one substring definition
    whose span crosses chunk borders; against the oracle, bit for bit."""
    import torch
    alpha = np.array([97, 98, 99], dtype=np.uint8)
    lines = ["0", str(S - 1), str(S - 1)] + ["%d %d %d" % (st, (st + 1 + i) % S, int(ch)) for st in range(S) for i, ch in enumerate(alpha)]
    pairs = sorted({(st, (st + 1 + i) % S) for st in range(S) for i in range(3)})
    sub = "\n".join(["8", "0", "99", "0 ", "%d " % (S - 1)] + ["%d %d" % p for p in pairs[: len(pairs) // 2]]) + "\n"
    text = "\n".join(lines) + "\n"
    B, M = 96, 8192
    rng = np.random.default_rng(S)
    chars = alpha[rng.integers(0, 3, size=(B, M))]
    lens = rng.integers(M // 2, M + 1, size=B).astype(np.uint32)
    lens[:8] = [M, M - 1, 1024, 1025, 1023, 0, 4096, 7 * 1024]
    cfg = hra.RegexVerifyConfig.configure(M, [hra.RegexDefs(hra.AllstrRegexDef(text), [hra.SubstrRegexDef(sub)])], device=0)
    assert "chunked=8x16 tiles" in cfg.describe_launch(B, layout=3)
    orec, omsk, ost = OracleDefs(oracle, [(text, [sub])]).witness_batch(chars, lens, M)
    dev = torch.device("cuda", 0)
    d_chars, d_lens = torch.from_numpy(chars).to(dev), torch.from_numpy(lens.astype(np.int32)).to(dev)
    rec, msk, st = cfg.witness_batch_position_major(hra.chars_to_position_major(d_chars), d_lens, chars_pm_stride=M)
    torch.cuda.synchronize()
    r1, m1 = hra.position_major_to_string_major(rec, msk, B, M, 1)
    assert np.array_equal(st.cpu().numpy().view(np.uint64), ost)
    assert np.array_equal(r1.cpu().numpy().view(np.uint32), orec) and np.array_equal(m1.cpu().numpy().view(np.uint16), omsk)


def test_three_defs_every_kernel_agrees_at_a_chip_filling_size(hra, oracle, monkeypatch):
    """D = 3, 16384 x 1024-byte strings (one group per CU): the def-parallel kernel (the planner's choice), the regular
    position-major kernel, its narrow-table build and the string-major path must produce the same bytes, and every string
    equals the oracle's."""
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
    orec, omsk, ost = OracleDefs.from_files(oracle, CFG_123).witness_batch(chars, lens, M, threads=os.cpu_count() or 1)
    assert np.array_equal(ref[2].cpu().numpy().view(np.uint64), ost)
    assert np.array_equal(ref[0].cpu().numpy().view(np.uint32), orec) and np.array_equal(ref[1].cpu().numpy().view(np.uint16), omsk)


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


HDR = [["header_from_lookup.txt", ["header_from_substr0.txt"]], ["header_to_lookup.txt", ["header_to_substr0.txt"]],
       ["header_subject_lookup.txt", ["header_subject_substr%d.txt" % k for k in range(3)]]]
CFG_D4 = CFG_123 + [HDR[0]]
CFG_D5 = [CFG_3[0], HDR[2], CFG_1[0], HDR[1], CFG_A[1]]
CFG_D7 = CFG_123 + [CFG_EX[0]] + HDR                    # def 3 = the partial example DFA: undefined transitions on almost any noise
CFG_D8 = CFG_123 + HDR + [CFG_1[0], CFG_A[1]]           # regex1 and regex2 a second time: both copies flag the same rows


CFG_D6 = HDR + CFG_123


@pytest.mark.parametrize("names", [CFG_D4, CFG_D5, CFG_D6, CFG_D7, CFG_D8], ids=["D4", "D5", "D6", "D7", "D8"])
def test_four_to_eight_defs_in_one_def_parallel_launch(hra, oracle, names, monkeypatch):
    """Configs of four to eight defs whose defs have at most 32 byte classes each are walked by ONE def-parallel launch on the CLASS-WIDE tables (hrx_kernel_pmd.hip CW: a walker wave per def
    over 256-byte rows behind the def's class LUT, a combiner wave of its own) instead of passes over groups of three.  Every
    string against the oracle: aligned and odd row counts, ragged and failing strings, n > M, both input layouts, string-major through the transposer — and a batch of 70000 strings, two
    blocks of the position-major buffers, in multi-round launches (the groups' block addressing)."""
    from halo2_regex_amd import synth
    D = len(names)
    for M in (328, 203):
        cfg = _cfg(hra, names, M)
        assert cfg.describe_launch(700, layout=3).startswith("hrx::witness_pmd_kernel<%d, true, true, false>" % D)
        chars, lens = synth.reveal_stress(500, min(M - 8, 320), seed=37)
        h_c, h_l = synth.headers_planted(200, chars.shape[1] - 3, seed=5, stride=chars.shape[1])
        chars, lens = np.concatenate([chars, h_c]), np.concatenate([lens, h_l])
        chars[7, 50] = 250                                       # a byte no DFA has a transition for
        chars[9, 0] = 200
        lens[11] = M + 72                                        # n > M
        lens[12], lens[13] = 0, min(M, chars.shape[1])
        _check_batch_pm(hra, oracle, names, chars, lens, M)                   # position-major outputs, both input layouts
        if M % 8 == 0:
            _check_batch(hra, oracle, names, chars, lens, M)                  # string-major: the same launch into scratch + the transposer
    # string-major rows in multiples of 16, four and five defs: the launch writes the caller's [B][pitch][D] records and [B][pitch] masked rows itself (storer wave, LDS sub-tiles)
    for M in (336, 64, 1024):
        cfg = _cfg(hra, names, M)
        d0 = cfg.describe_launch(700, layout=0)
        assert d0.startswith("hrx::witness_pmd_kernel<%d, true, true, true>" % D) == (D <= 5) and ("transpose_pm_to_sm_kernel" in d0) == (D > 5)
        chars, lens = synth.reveal_stress(700, M - 1, seed=43)
        h_c, h_l = synth.headers_planted(200, min(chars.shape[1] - 3, M - 1), seed=7, stride=chars.shape[1])
        chars, lens = np.concatenate([chars, h_c]), np.concatenate([lens, h_l])
        chars[5, 20] = 250
        lens[11] = M + 72
        lens[12], lens[13] = 0, min(M, chars.shape[1])
        _check_batch(hra, oracle, names, chars, lens, M)
        if M == 64:            # batches smaller than a group, and a group plus one string
            for Bs in (1, 65):
                _check_batch(hra, oracle, names, np.ascontiguousarray(chars[100:100 + Bs]), lens[100:100 + Bs].copy(), M)
        if M == 1024:          # device-resident, pitched string-major buffers (hrx_recommended_pitches), several rounds of groups
            import torch
            dev = torch.device("cuda", 0)
            big_c, big_l = np.tile(chars, (20, 1)), np.tile(lens, 20)
            o = OracleDefs.from_files(oracle, names)
            orec, omsk, ost = o.witness_batch(big_c, big_l, M, threads=os.cpu_count() or 8)
            wide = torch.zeros((len(big_l), (big_c.shape[1] + 15) // 16 * 16), dtype=torch.uint8, device=dev)
            wide[:, :big_c.shape[1]] = torch.from_numpy(big_c).to(dev)
            out = cfg.alloc_outputs(len(big_l), dev, pitched=True)
            rec, msk, st = cfg.witness_batch(wide, torch.from_numpy(big_l.astype(np.int32)).to(dev), out=out)
            torch.cuda.synchronize()
            ok = (ost & np.uint64(0xff)) == 0
            assert np.array_equal(st.cpu().numpy().view(np.uint64), ost)
            assert np.array_equal(rec.cpu().numpy().view(np.uint32)[ok], orec[ok]) and np.array_equal(msk.cpu().numpy().view(np.uint16)[ok], omsk[ok])
    M = 136
    B = 70000
    chars, lens = synth.ragged(B, M, seed=29)
    cfg = _cfg(hra, names, M)
    assert cfg.describe_launch(B, layout=3).startswith("hrx::witness_pmd_kernel<%d, true, true, false>" % D)
    blocks = [(chars[:hra.PM_BLOCK], lens[:hra.PM_BLOCK]), (np.ascontiguousarray(chars[hra.PM_BLOCK:]), lens[hra.PM_BLOCK:])]
    st = _full_check(hra, OracleDefs.from_files(oracle, names), cfg, blocks, M, D)
    # (D7's partial example DFA fails on almost any noise, D8's two copies of regex1 / regex2 flag the same rows: out of contract — their status words are the oracle's too)
    assert len(st) == B and (names is CFG_D7 or names is CFG_D8 or (st & np.uint64(0xff) == 0).mean() > 0.5)


@pytest.mark.parametrize("combine", [False, True], ids=["merged-by-the-last-pass", "combine-launch"])
@pytest.mark.parametrize("names", [CFG_D4, CFG_D5, CFG_D7, CFG_D8], ids=["D4", "D5", "D7", "D8"])
def test_more_than_three_regex_defs_multi_pass(hra, oracle, names, combine, monkeypatch):
    """regex_defs is a Vec of any length (src/lib.rs:112): more than three defs are walked in passes (groups of defs); what needs all
    defs of a row is formed by the LAST pass from the earlier groups' tile summaries (position-major outputs, up to four groups) or by
    hrx::witness_combine_kernel / _summary_kernel (HRX_MP_COMBINE=1 forces it) — every layout (string-major, position-major,
    position-major input), substr ids counting on across the groups (lib.rs:827,842), merged status words: the lowest def's undefined
    transition, the flag-overlap row, the accept mask with a bit per def; then every ok string through the integer MockProver."""
    import torch
    from halo2_regex_amd import synth
    from mock_prover import IntegerMockProver, FAIL_ACCEPT
    D = len(names)
    monkeypatch.setenv("HRX_MP_COMBINE", "1" if combine else "0")
    for M in (328, 203):                                       # aligned and unaligned row counts
        cfg = _cfg(hra, names, M)
        d = cfg.describe_launch(700, layout=3)
        # (four to seven defs of at most 32 byte classes each, without HRX_MP_COMBINE: ONE def-parallel launch on the CLASS-WIDE tables instead of passes — D7's partial example DFA included;
        # with HRX_MP_COMBINE=1, and for D8, the passes over groups of three defs)
        assert (d.startswith("multi-pass, ") and ("witness_combine_summary_kernel" if combine else "witness_merge_status_kernel") in d) or \
            (not combine and D in (4, 5, 6, 7, 8) and d.startswith("hrx::witness_pmd_kernel<%d, true, true, false>" % D))
        chars, lens = synth.reveal_stress(500, min(M - 8, 320), seed=31)
        h_c, h_l = synth.headers_planted(200, chars.shape[1] - 3, seed=3, stride=chars.shape[1])
        chars, lens = np.concatenate([chars, h_c]), np.concatenate([lens, h_l])
        chars[7, 50] = 250                                       # a byte no DFA has a transition for
        lens[11] = M + 72                                        # n > M
        lens[12], lens[13] = 0, min(M, chars.shape[1])
        st, _ = _check_batch(hra, oracle, names, chars, lens, M)              # string-major (host buffers -> device batch kernels)
        _check_batch_pm(hra, oracle, names, chars, lens, M)                   # position-major outputs, both input layouts
        codes = st & np.uint64(0xff)
        assert (codes == 1).any() and (codes == 3).any()
        if names is CFG_D7:
            assert ((st[codes == 1] >> np.uint64(8)) & np.uint64(0xff) == 3).sum() > 300
        elif names is CFG_D8:
            assert (codes == 2).sum() > 300
        else:
            assert (codes == 0).sum() > 100 and len(set(int(x) for x in st[codes == 0] >> np.uint64(8))) > 2
        # the constraint system on the device rows
        dev = torch.device("cuda", 0)
        d_chars, d_lens = torch.from_numpy(chars).to(dev), torch.from_numpy(lens.astype(np.int32)).to(dev)
        rec, msk, dst = cfg.witness_batch(d_chars, d_lens)
        torch.cuda.synchronize()
        code = IntegerMockProver.from_config(cfg, device=dev).verify(d_chars, d_lens, rec, msk, M).cpu().numpy()
        ok = codes == 0
        accepted = ((st >> np.uint64(8)) == np.uint64((1 << D) - 1)) | (lens >= M)
        assert (code[ok & accepted] == 0).all() and (code[ok & ~accepted] == FAIL_ACCEPT).all()


NOSUB = lambda cfg: [[a, []] for a, _ in cfg]                # the same DFAs without substring definitions: walked for acceptance only (no flags, no ids)
CFG_D13 = CFG_123 + HDR + NOSUB(CFG_123 + HDR) + [CFG_EX[0]]                          # five groups: past what the last pass merges itself
CFG_D16 = HDR + NOSUB(CFG_123) + CFG_123 + NOSUB(HDR + CFG_123) + [CFG_1[0]]          # def 15 = regex1 + substr1 again (also def 6): flags of groups 2 and 5 overlap
CFG_D32 = CFG_123 + NOSUB(HDR + CFG_123) * 4 + HDR + NOSUB(CFG_A)                     # HRX_MAX_DEFS: accept bits 8 .. 39


@pytest.mark.parametrize("combine", [False, True], ids=["default", "combine-launch"])
@pytest.mark.parametrize("names", [CFG_D13, CFG_D16, CFG_D32], ids=["D13", "D16", "D32"])
def test_up_to_32_regex_defs(hra, oracle, names, combine, monkeypatch):
    """HRX_MAX_DEFS = 32 RegexDefs per config (the status word's accept mask, bits 8..39; include/hrx.h): 13, 16 and 32 defs = 2 .. 4 CW groups of up to eight defs (position-major passes: one
    def-parallel launch each) or 5 .. 11 groups of three — more than the last pass merges itself, so the combine launch forms what needs all defs of a row whatever HRX_MP_COMBINE says — string-major outputs (through the position-major path
    and the transpose kernel for row counts in multiples of 8: 16-row tiles from 10 defs on, hrx_kernel_tp.hip; the copy-mode combine otherwise), position-major outputs
    with both input layouts; accept masks with a bit per def, the lowest def's undefined transition, flags of defs in different groups on one row."""
    from halo2_regex_amd import synth
    D = len(names)
    assert D == {id(CFG_D13): 13, id(CFG_D16): 16, id(CFG_D32): 32}[id(names)]
    monkeypatch.setenv("HRX_MP_COMBINE", "1" if combine else "0")
    for M in (328, 203):                                       # aligned (transpose kernel) and unaligned (copy-mode combine) row counts
        cfg = _cfg(hra, names, M)
        d = cfg.describe_launch(700, layout=3)
        # (every def of these configs has at most 32 byte classes: passes over CW groups of up to eight defs — 2, 2 and 4 def-parallel launches; the groups of three, 5 .. 11 passes, serve the
        # unaligned string-major row counts below and kDbgNoDefParallel)
        assert d.startswith("multi-pass, ") and "witness_combine_summary_kernel" in d and d.count("[defs ") >= 2 and "witness_pmd_kernel<" in d
        assert ("transpose_pm_to_sm_kernel" in cfg.describe_launch(700, layout=0)) == (M % 8 == 0)
        chars, lens = synth.reveal_stress(500, min(M - 8, 320), seed=41)
        h_c, h_l = synth.headers_planted(200, chars.shape[1] - 3, seed=5, stride=chars.shape[1])
        chars, lens = np.concatenate([chars, h_c]), np.concatenate([lens, h_l])
        if names is not CFG_D13:
            chars[7, 50] = 250                                   # a byte no DFA has a transition for
        lens[11] = M + 72                                        # n > M
        lens[12], lens[13] = 0, min(M, chars.shape[1])
        st, _ = _check_batch(hra, oracle, names, chars, lens, M)              # string-major (host buffers -> device batch kernels)
        _check_batch_pm(hra, oracle, names, chars, lens, M)                   # position-major outputs, both input layouts
        codes = st & np.uint64(0xff)
        assert (codes == 3).any()
        if names is CFG_D13:
            assert ((st[codes == 1] >> np.uint64(8)) & np.uint64(0xff) == 12).sum() > 300     # def 12 = the partial example DFA
        elif names is CFG_D16:
            assert (codes == 2).sum() > 100                      # regex1 + substr1 as def 6 and def 15
        else:
            masks = set(int(x) for x in st[codes == 0] >> np.uint64(8))
            assert (codes == 0).sum() > 100 and len(masks) > 2 and max(masks) >= 1 << 24      # accept bits of defs beyond 24


@pytest.mark.parametrize("big_first", [True, False], ids=["big-def-first", "big-def-last"])
@pytest.mark.parametrize("combine", [False, True], ids=["merged-by-the-last-pass", "combine-launch"])
def test_multi_pass_with_a_group_on_the_byte_table(hra, oracle, big_first, combine, monkeypatch):
    """A config of four defs one of which is a 200-state DFA: that def is a group of its own whose 4-byte table does not fit LDS — its pass runs on the BYTE table
    (tag bytes -> bitvectors in the finisher, hrx_kernel_pm.hip byte_tile_bits) and either writes the tile summaries the other passes' merge reads (big def first)
    or is the LAST pass that merges the earlier groups' summaries itself (big def last); both output layouts, every string against the oracle."""
    import torch
    from halo2_regex_amd import synth
    monkeypatch.setenv("HRX_MP_COMBINE", "1" if combine else "0")
    read = lambda f: open(os.path.join(DFA_DIR, f)).read()
    small = [(read(a), [read(x) for x in subs]) for a, subs in CFG_123]
    big_a, big_s = synth.random_dfa(200, seed=12, n_substr_pairs=60)
    defs_t = ([(big_a, [big_s])] + small) if big_first else (small + [(big_a, [big_s])])
    M, B = 328, 700
    chars, lens = synth.reveal_stress(B - 200, M - 8, seed=51)
    n_c, n_l = synth.noise(200, M - 8, seed=2, stride=chars.shape[1])
    chars, lens = np.concatenate([chars, n_c]), np.concatenate([lens, n_l])
    lens[5], lens[6] = 0, M
    defs = [hra.RegexDefs(hra.AllstrRegexDef(a), [hra.SubstrRegexDef(t) for t in subs]) for a, subs in defs_t]
    cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
    d = cfg.describe_launch(B, layout=3)
    assert d.startswith("multi-pass, 2 groups") and "witness_pm_kernel<1, false, false, false, false, true>" in d     # the big def's group: the BYTE table
    o = OracleDefs(oracle, defs_t)
    orec, omsk, ost = o.witness_batch(chars, lens, M)
    ok = (ost & np.uint64(0xff)) == 0
    assert ok.sum() > 100
    grec, gmsk, gst = cfg.witness_batch_host(chars, lens)                   # string-major outputs (position-major passes + transpose)
    assert np.array_equal(ost, gst) and np.array_equal(orec[ok], grec[ok]) and np.array_equal(omsk[ok], gmsk[ok])
    dev = torch.device("cuda", 0)
    d_chars, d_lens = torch.from_numpy(chars).to(dev), torch.from_numpy(lens.astype(np.int32)).to(dev)
    rec, msk, st = cfg.witness_batch_position_major(hra.chars_to_position_major(d_chars), d_lens, chars_pm_stride=chars.shape[1])
    torch.cuda.synchronize()
    r1, m1 = hra.position_major_to_string_major(rec, msk, B, M, 4)
    assert np.array_equal(st.cpu().numpy().view(np.uint64), ost)
    assert np.array_equal(r1.cpu().numpy().view(np.uint32)[ok], orec[ok]) and np.array_equal(m1.cpu().numpy().view(np.uint16)[ok], omsk[ok])


def test_multi_pass_at_a_chip_filling_size_two_blocks(hra, oracle):
    """D = 5 at 70000 x 1023 bytes (two blocks of the position-major buffers): every string against the oracle and the MockProver."""
    from halo2_regex_amd import synth
    M, n = 1024, 1023
    base_c, base_l = synth.regex23_planted(hra.PM_BLOCK, n, seed=9, stride=1024)
    blocks = _rolled_blocks(base_c, base_l, 2, last=70000 - hra.PM_BLOCK, seed=6)
    cfg = _cfg(hra, CFG_D5, M)
    st = _full_check(hra, OracleDefs.from_files(oracle, CFG_D5), cfg, blocks, M, 5)
    assert len(st) == 70000 and (st & np.uint64(0xff) == 0).mean() > 0.9
    _full_check(hra, OracleDefs.from_files(oracle, CFG_D5), cfg, [(base_c[:5000], base_l[:5000])], M, 5, position_major=False)


@pytest.mark.parametrize("flags,names", [(0x1000, CFG_1), (0x1000, CFG_A), (0x1000 | 0x400000, CFG_123), (0x1000 | 0x2000 | 0x8000000, CFG_1)],
                         ids=["D1", "D2-wide", "D3-half", "D1-byte"])
def test_dynamic_group_assignment(hra, oracle, flags, names, monkeypatch):
    """Batches of eight or more long groups per walker pair take their groups from a device counter instead of a fixed stride (the
    loader draws, walker and finisher follow through an LDS queue).  Forced here from the second group on (kDbgForceDynamicGroups)
    at 70000 ragged strings — 1094 groups over 1024 pairs, two blocks of the position-major buffers, three tiles per group, one of
    them partial — and replayed from a HIP graph (the counter is reset by a memset node in front of every launch): every string
    against the oracle and the MockProver, for the finisher variants and the HALF one."""
    import torch
    from halo2_regex_amd import synth
    monkeypatch.setenv("HRX_DEBUG_FLAGS", _flags(flags))
    M, B = 136, 70000
    chars, lens = synth.ragged(B, M, seed=23)
    cfg = _cfg(hra, names, M)
    assert cfg.describe_launch(B, layout=3).endswith("groups=dynamic")
    blocks = [(chars[:hra.PM_BLOCK], lens[:hra.PM_BLOCK]), (np.ascontiguousarray(chars[hra.PM_BLOCK:]), lens[hra.PM_BLOCK:])]
    st = _full_check(hra, OracleDefs.from_files(oracle, names), cfg, blocks, M, len(names))
    assert (st & np.uint64(0xff) == 0).mean() > 0.9
    # graph replay: four captured launches INTO FOUR DIFFERENT OUTPUT SETS, replayed three times, every set the same bytes as an eager launch
    # (with one set a launch that skipped its dynamic groups goes unnoticed behind the launches around it: round 3 found the counter's
    # reset — then a memset node — executing out of order with the kernel nodes from the second replay on)
    dev = torch.device("cuda", 0)
    d_chars, d_lens = torch.from_numpy(chars).to(dev), torch.from_numpy(lens.astype(np.int32)).to(dev)
    ref = cfg.alloc_outputs_position_major(B, dev)
    for t in ref:
        t.fill_(-1)          # (cells the contract leaves unspecified — rows of a string whose status is not 0 — keep the fill on both sides)
    cfg.witness_batch_position_major(d_chars, d_lens, out=ref)
    outs = [cfg.alloc_outputs_position_major(B, dev) for _ in range(4)]
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            for out in outs:
                cfg.witness_batch_position_major(d_chars, d_lens, out=out)
    torch.cuda.current_stream(dev).wait_stream(side)
    for _ in range(3):
        for out in outs:
            for t in out:
                t.fill_(-1)
        g.replay()
        torch.cuda.synchronize()
        for out in outs:
            assert all(torch.equal(x, y) for x, y in zip(out, ref))


def test_placement_aware_output_allocation(hra, oracle):
    """hrx_alloc_outputs_position_major / hrx_device_free (include/hrx.h): below 1 GiB of records two plain allocations, from 1 GiB
    on the masked-row buffer is the best of several candidates measured against the records buffer; either way ordinary device
    memory that a launch fills like any other (here 140000 x 1024 rows at D = 2, 1.07 GiB of records: every string against the oracle)."""
    import ctypes as C
    import torch
    from halo2_regex_amd import synth
    dev = torch.device("cuda", 0)
    M, B = 1024, 140000
    cfg = _cfg(hra, CFG_A, M)
    base_c, base_l = synth.reveal_stress(hra.PM_BLOCK, M - 7, seed=31)
    blocks = _rolled_blocks(base_c, base_l, 3, last=B - 2 * hra.PM_BLOCK, seed=8)
    free0 = torch.cuda.mem_get_info()[0]
    out = cfg.alloc_outputs_position_major(B, dev)
    assert out[0].numel() * 4 >= hra.PLACED_FROM and out[0].data_ptr() % 16 == 0 and out[1].data_ptr() % 16 == 0
    used = free0 - torch.cuda.mem_get_info()[0]
    assert used < out[0].numel() * 4 + out[1].numel() * 2 + (256 << 20)        # the losing candidates and the spacers were freed
    rep = cfg.last_placement_report()
    assert rep["searched"] == 1 and rep["steps"] >= 1 and rep["ref_us"] > 0 and rep["best_us"] > 0 and rep["best_us"] <= rep["first_us"]
    assert out[1].numel() * 2 <= rep["peak_candidate_bytes"] <= 0.70 * free0   # bounded: never more than 70 % of the free memory
    del out
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info()[0] < (256 << 20)                    # ... and hrx_device_free released the pair
    st = _full_check(hra, OracleDefs.from_files(oracle, CFG_A), cfg, blocks, M, 2)      # (its launches allocate their outputs the same way)
    assert len(st) == B
    pr, pm = C.c_void_p(), C.c_void_p()
    assert hra.lib.hrx_alloc_outputs_position_major(cfg._ctx, 1000, 64, C.byref(pr), C.byref(pm)) == hra.HRX_OK and pr.value and pm.value
    assert hra.lib.hrx_device_free(pr) == hra.HRX_OK and hra.lib.hrx_device_free(pm) == hra.HRX_OK and hra.lib.hrx_device_free(None) == hra.HRX_OK
    assert cfg.last_placement_report()["searched"] == 0                          # a launch that lives in the Infinity Cache: two plain allocations
    assert hra.lib.hrx_alloc_outputs_position_major(cfg._ctx, 0, 64, C.byref(pr), C.byref(pm)) == hra.HRX_ERR_ARG
    assert hra.lib.hrx_alloc_outputs_position_major(None, 8, 64, C.byref(pr), C.byref(pm)) == hra.HRX_ERR_ARG
    # more than the device holds: a loud error, nothing left allocated (records too big; records fit but no masked-row candidate does)
    free1 = torch.cuda.mem_get_info()[0]
    assert hra.lib.hrx_alloc_output_pair(cfg._ctx, 1 << 40, 1 << 20, C.byref(pr), C.byref(pm)) == hra.HRX_ERR_HIP and not pr.value and not pm.value
    assert hra.lib.hrx_alloc_output_pair(cfg._ctx, 2 << 30, 1 << 40, C.byref(pr), C.byref(pm)) == hra.HRX_ERR_HIP and not pr.value and not pm.value
    assert abs(free1 - torch.cuda.mem_get_info()[0]) < (64 << 20)


def test_placement_search_for_bench_sized_buffers_uses_measured_arenas(hra, oracle, monkeypatch):
    """Records below 1 GiB (the bench line: 256 MiB of records, 128 MiB of masked rows) are carved out of a measured pair of 2-GiB
    arenas — a probe over buffers that fit the Infinity Cache would measure the cache: the first call walks (never more than 70 %
    of the free memory), later calls are served from the same pair, a full pair is replaced, hrx_device_free returns sub-buffers
    and the arenas go with the device's last context, and the buffers are ordinary memory (one launch against the oracle, every string).
    HRX_PLACE=0 (read at context creation) turns the search off."""
    import torch
    from halo2_regex_amd import synth
    dev = torch.device("cuda", 0)
    M, B = 1024, 65536
    import gc
    gc.collect()                        # (the arena pair belongs to the device: no context of an earlier test may still hold it)
    cfg = _cfg(hra, CFG_1, M)
    torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    outs = [cfg.alloc_outputs_position_major(B, dev)]
    rep = cfg.last_placement_report()
    nrec, nmsk = outs[0][0].numel() * 4, outs[0][1].numel() * 2
    assert nrec == 256 << 20 and rep["searched"] == 1 and rep["steps"] >= 1 and rep["ref_us"] > 0 and 0 < rep["best_us"] <= rep["first_us"]
    assert (2 << 30) <= rep["peak_candidate_bytes"] <= 0.70 * free0 and rep["probe_bytes"] >= 1 << 30
    assert free0 - torch.cuda.mem_get_info()[0] < (4 << 30) + (256 << 20)            # the pair of arenas, nothing else
    for k in range(7):                                                                # 8 x 256 MiB fill the records arena exactly
        outs.append(cfg.alloc_outputs_position_major(B, dev))
        assert cfg.last_placement_report()["searched"] == 2
    ptrs = sorted(o[0].data_ptr() for o in outs)
    assert all(b - a == nrec for a, b in zip(ptrs, ptrs[1:]))                         # back to back inside one 2-GiB block
    assert free0 - torch.cuda.mem_get_info()[0] < (4 << 30) + (256 << 20)
    outs.append(cfg.alloc_outputs_position_major(B, dev))                             # the pair is full: a new one is measured
    assert cfg.last_placement_report()["searched"] == 1
    chars, lens = synth.reveal_stress(B, M - 1, seed=5)
    d_c, d_l = hra.chars_to_position_major(torch.from_numpy(chars).to(dev)), torch.from_numpy(lens.astype(np.int32)).to(dev)
    rec, msk, st = cfg.witness_batch_position_major(d_c, d_l, out=outs[3], chars_pm_stride=chars.shape[1])
    torch.cuda.synchronize()
    orec, omsk, ost = OracleDefs.from_files(oracle, CFG_1).witness_batch(chars, lens, M, threads=os.cpu_count() or 8)
    r2, m2 = hra.position_major_to_string_major(rec, msk, B, M, 1)
    ok = torch.from_numpy((ost & np.uint64(0xff)) == 0).to(dev)
    assert np.array_equal(st.cpu().numpy().view(np.uint64), ost)
    assert torch.equal(r2[ok], torch.from_numpy(orec.view(np.int32)).to(dev)[ok]) and torch.equal(m2[ok], torch.from_numpy(omsk.view(np.int16)).to(dev)[ok])
    # hrx_traffic_pass_device: the launch's memory traffic with no DFA work — it overwrites these outputs (with junk) and nothing else
    before = [o[0][:4096].clone() for o in outs]
    cfg.traffic_pass(d_c, B, outs[3], chars.shape[1])
    torch.cuda.synchronize()
    r3, _ = hra.position_major_to_string_major(outs[3][0], outs[3][1], B, M, 1)
    assert not torch.equal(r3[ok], torch.from_numpy(orec.view(np.int32)).to(dev)[ok])
    assert all(torch.equal(o[0][:4096], b) for k, (o, b) in enumerate(zip(outs, before)) if k != 3)
    del outs, rec, msk, st, r2, m2, r3, d_c, d_l, ok, before
    del cfg                                                                           # the device's last context goes: the arenas are released
    import gc
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()       # (the test's own tensors sit in torch's caching allocator)
    assert free0 - torch.cuda.mem_get_info()[0] < (256 << 20)
    monkeypatch.setenv("HRX_PLACE", "0")
    cfg0 = _cfg(hra, CFG_1, M + 8)
    out = cfg0.alloc_outputs_position_major(B, dev)
    assert cfg0.last_placement_report()["searched"] == 0 and out[0].numel() * 4 == (M + 8) * B * 4


def test_placement_arenas_are_shared_by_the_contexts_of_a_device(hra, oracle):
    """One measured arena pair per device and process: a second context (another config: one context per worker thread is how a prover gets overlap) is served from
    the pair the first one walked for — no second walk, no second 4 GiB — the pair outlives the context that measured it while another context of the device lives,
    and goes with the last one."""
    import gc
    import torch
    dev = torch.device("cuda", 0)
    M, B = 1024, 65536
    gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    a = _cfg(hra, CFG_1, M)
    out_a = a.alloc_outputs_position_major(B, dev)
    assert a.last_placement_report()["searched"] == 1
    b = _cfg(hra, CFG_23, M)
    out_b = b.alloc_outputs_position_major(B, dev)
    rep_b = b.last_placement_report()
    assert rep_b["searched"] == 2 and rep_b["steps"] == a.last_placement_report()["steps"]           # the first context's walk, reported again
    assert 0 < out_b[0].data_ptr() - out_a[0].data_ptr() <= (2 << 30) - (512 << 20)                   # inside the same 2-GiB records arena
    assert free0 - torch.cuda.mem_get_info()[0] < (4 << 30) + (256 << 20)                             # ONE pair of arenas for both contexts
    del a, out_a
    gc.collect()
    out_b2 = b.alloc_outputs_position_major(B, dev)                                                   # the pair outlives the context that measured it
    assert b.last_placement_report()["searched"] == 2
    from halo2_regex_amd import synth
    chars, lens = synth.reveal_stress(B, M - 1, seed=11)
    d_c, d_l = hra.chars_to_position_major(torch.from_numpy(chars).to(dev)), torch.from_numpy(lens.astype(np.int32)).to(dev)
    rec, msk, st = b.witness_batch_position_major(d_c, d_l, out=out_b2, chars_pm_stride=chars.shape[1])
    torch.cuda.synchronize()
    orec, omsk, ost = OracleDefs.from_files(oracle, CFG_23).witness_batch(chars, lens, M, threads=os.cpu_count() or 8)
    r2, m2 = hra.position_major_to_string_major(rec, msk, B, M, 2)
    ok = torch.from_numpy((ost & np.uint64(0xff)) == 0).to(dev)
    assert np.array_equal(st.cpu().numpy().view(np.uint64), ost)
    assert torch.equal(r2[ok], torch.from_numpy(orec.view(np.int32)).to(dev)[ok]) and torch.equal(m2[ok], torch.from_numpy(omsk.view(np.int16)).to(dev)[ok])
    del b, out_b, out_b2, rec, msk, st, r2, m2, d_c, d_l, ok
    gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
    assert free0 - torch.cuda.mem_get_info()[0] < (256 << 20)                                          # the last context of the device is gone: so is the pair


def test_host_buffer_batches_are_pipelined_chunk_by_chunk(hra, oracle):
    """hrx_witness_batch_host — what an unmodified caller of the seam gets (host Vecs in, host Vecs out, lib.rs:311-318) — stages, walks and copies a large batch out
    chunk by chunk on two streams and two host threads, or in one piece on one stream, whichever the context finds faster on the box.  Every string against the oracle: a batch of several chunks with a last partial one, ragged strings, strings
    with undefined transitions in several chunks, a stride that is not a multiple of 16 (the staging pads it), two defs; and the same rows gathered per circuit out of
    the position-major device buffers copied to the host as they are (hrx_rows_of_string_position_major)."""
    import torch
    from halo2_regex_amd import synth
    M = 1024
    for names, B, stride in ((CFG_1, 40000, 1024), (CFG_A, 21001, 1023)):
        D = len(names)
        chars, lens = synth.regex1_planted(B, min(stride, M - 1), seed=41, stride=stride) if stride % 16 == 0 else synth.regex1_planted(B, 1008, seed=41, stride=1008)
        if chars.shape[1] != stride:                       # an odd stride: re-pack the strings 1023 bytes apart
            wide = np.zeros((B, stride), np.uint8); wide[:, :chars.shape[1]] = chars; chars = wide
        rng = np.random.default_rng(3)
        idx = rng.choice(B, 300, replace=False)
        lens[idx[:200]] = rng.integers(0, lens[idx[:200]] + 1)
        chars[idx[200:], rng.integers(0, 900, 100)] = 200   # a byte no definition has a transition for: lib.rs:817 in chunks all over the batch
        cfg = _cfg(hra, names, M)
        grec, gmsk, gst = cfg.witness_batch_host(chars, lens)
        o = OracleDefs.from_files(oracle, names)
        orec, omsk, ost = o.witness_batch(chars, lens, M, threads=os.cpu_count() or 8)
        ok = (ost & np.uint64(0xff)) == 0
        assert np.array_equal(gst, ost) and (~ok).sum() >= 90
        assert np.array_equal(grec[ok], orec[ok]) and np.array_equal(gmsk[ok], omsk[ok])
        # the context compares its two ways over its next calls (pipelined, one stream, pipelined, one stream: hrx_api.cpp batch_host_locked) and then keeps the faster one: the same rows every time
        for call in range(5):
            r2, m2, s2 = cfg.witness_batch_host(chars, lens)
            assert np.array_equal(s2, ost) and np.array_equal(r2[ok], orec[ok]) and np.array_equal(m2[ok], omsk[ok]), call
        # per-circuit view of the position-major buffers, on the host
        dev = torch.device("cuda", 0)
        padded = chars if stride % 16 == 0 else np.pad(chars, ((0, 0), (0, 16 - stride % 16)))
        d_c = hra.chars_to_position_major(torch.from_numpy(padded).to(dev))
        rec, msk, st = cfg.witness_batch_position_major(d_c, torch.from_numpy(lens.astype(np.int32)).to(dev), chars_pm_stride=padded.shape[1])
        torch.cuda.synchronize()
        rp, mp = rec.cpu().numpy().view(np.uint32), msk.cpu().numpy().view(np.uint16)
        for b in [0, B - 1] + [int(x) for x in rng.choice(np.nonzero(ok)[0], 40)]:
            if not ok[b]:
                continue
            r1, m1 = hra.rows_of_string_position_major(rp, mp, B, M, D, b)
            assert np.array_equal(r1, orec[b]) and np.array_equal(m1, omsk[b]), b


def test_context_clones_run_side_by_side(hra, oracle):
    """hrx_ctx_clone: what `impl Clone for RegexVerifyConfig` maps to — two clones of one config launch on two streams from two threads at the same time (a context serves
    one stream at a time; its clones have their own scratch and lock) and write the oracle's rows; a host-only clone of a device context serves the single-string seam."""
    import threading
    import torch
    from halo2_regex_amd import synth
    dev = torch.device("cuda", 0)
    M, B = 2048, 70000
    chars, lens = synth.regex23_planted(4096, M - 1, seed=5, stride=M)
    chars, lens = np.tile(chars, (18, 1))[:B], np.tile(lens, 18)[:B]
    cfg = _cfg(hra, CFG_23, M)
    clones = [cfg.clone(), cfg.clone()]
    d_c = hra.chars_to_position_major(torch.from_numpy(chars).to(dev))
    d_l = torch.from_numpy(lens.astype(np.int32)).to(dev)
    torch.cuda.synchronize()
    outs, errs = [None, None], []

    def work(k):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                for _ in range(3):
                    outs[k] = clones[k].witness_batch_position_major(d_c, d_l, chars_pm_stride=M)      # multi-round: dynamic groups, the context's counter
                torch.cuda.current_stream().synchronize()
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errs, errs
    ref = cfg.witness_batch_position_major(d_c, d_l, chars_pm_stride=M)
    torch.cuda.synchronize()
    for k in range(2):
        assert all(torch.equal(x, y) for x, y in zip(outs[k], ref))
    o = OracleDefs.from_files(oracle, CFG_23)
    orec, omsk, ost = o.witness_batch(chars[:4096], lens[:4096], M, threads=os.cpu_count() or 8)
    r2, m2 = hra.position_major_to_string_major(ref[0], ref[1], B, M, 2)
    assert np.array_equal(ref[2][:4096].cpu().numpy().view(np.uint64), ost)
    assert np.array_equal(r2[:4096].cpu().numpy().view(np.uint32), orec) and np.array_equal(m2[:4096].cpu().numpy().view(np.uint16), omsk)
    host = cfg.clone(device=hra.HRX_DEVICE_NONE)
    one = host.match_substrs(bytes(chars[0, :lens[0]]))
    assert np.array_equal(one.masked_characters.astype(np.uint16) | (one.all_substr_ids.astype(np.uint16) << 8), omsk[0])


def test_freed_arena_range_is_not_reused_before_the_device_has_drained(hra, oracle):
    """hrx_device_free on a sub-buffer of the shared arena pair (include/hrx.h): the range is not handed out again before the device has drained, so a buffer freed while
    its launch is still in flight never reaches a second context that writes it on another stream.  Context a queues launches into arena buffers on its own stream and frees
    them WITHOUT syncing: the free returns at once (the range is parked); context b allocates until the arena has nothing else left — the allocation that takes the parked
    range back has waited for the device — and its rows, written on another stream, are the oracle's."""
    import gc
    import torch
    from halo2_regex_amd import synth
    dev = torch.device("cuda", 0)
    M, B = 1024, 65536
    chars, lens = synth.reveal_stress(B, M - 1, seed=31)
    d_c, d_l = hra.chars_to_position_major(torch.from_numpy(chars).to(dev)), torch.from_numpy(lens.astype(np.int32)).to(dev)
    a, b = _cfg(hra, CFG_1, M), _cfg(hra, CFG_1, M)
    keep = a.alloc_outputs_position_major(B, dev)                      # (keeps the arena pair alive and a's context busy below)
    out_a = a.alloc_outputs_position_major(B, dev)
    assert a.last_placement_report()["searched"] in (1, 2)
    ptr_a = (out_a[0].data_ptr(), out_a[1].data_ptr())
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        for _ in range(40):                                             # ~3 ms of queued work writing out_a
            a.witness_batch_position_major(d_c, d_l, out=out_a, chars_pm_stride=chars.shape[1])
    assert not s1.query()                                               # still in flight
    del out_a
    gc.collect()                                                        # -> hrx_device_free of both sub-buffers, no sync by the caller
    held, out_b = [], None
    for _ in range(16):                                                 # (a 2-GiB arena has eight 256-MiB ranges)
        o = b.alloc_outputs_position_major(B, dev)
        if o[0].data_ptr() == ptr_a[0]:
            out_b = o
            break
        assert s1.query() or o[0].data_ptr() != ptr_a[0]
        held.append(o)
    assert out_b is not None, "the freed range never came back: the arena pair was replaced instead"
    assert s1.query(), "a freed arena range was handed out while launches into it were still running"
    del held
    with torch.cuda.stream(s2):
        rec, msk, st = b.witness_batch_position_major(d_c, d_l, out=out_b, chars_pm_stride=chars.shape[1])
    torch.cuda.synchronize()
    orec, omsk, ost = OracleDefs.from_files(oracle, CFG_1).witness_batch(chars, lens, M, threads=os.cpu_count() or 8)
    r2, m2 = hra.position_major_to_string_major(rec, msk, B, M, 1)
    ok = torch.from_numpy((ost & np.uint64(0xff)) == 0).to(dev)
    assert np.array_equal(st.cpu().numpy().view(np.uint64), ost)
    assert torch.equal(r2[ok], torch.from_numpy(orec.view(np.int32)).to(dev)[ok]) and torch.equal(m2[ok], torch.from_numpy(omsk.view(np.int16)).to(dev)[ok])
    del keep


def test_arena_free_does_not_disturb_a_capture_in_another_thread(hra, oracle):
    """While any thread captures a stream in the global capture mode (torch.cuda.graph's default) the runtime refuses a device-wide wait — and in the relaxed mode the wait
    invalidates that capture: hrx_device_free of an arena sub-buffer from another thread therefore does not wait at all (the range is parked), and the capture survives it."""
    import gc
    import threading
    import torch
    from halo2_regex_amd import synth
    dev = torch.device("cuda", 0)
    M, B = 1024, 65536
    chars, lens = synth.reveal_stress(B, M - 1, seed=33)
    d_c, d_l = hra.chars_to_position_major(torch.from_numpy(chars).to(dev)), torch.from_numpy(lens.astype(np.int32)).to(dev)
    a, c = _cfg(hra, CFG_1, M), _cfg(hra, CFG_1, M)
    keep = a.alloc_outputs_position_major(B, dev)
    out_a = a.alloc_outputs_position_major(B, dev)
    out_c = c.alloc_outputs_position_major(B, dev)
    s1 = torch.cuda.Stream(device=dev)
    inside, freed, result = threading.Event(), threading.Event(), {}

    def capturer():
        try:
            torch.cuda.set_device(0)
            side = torch.cuda.Stream(device=dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side, capture_error_mode="global"):
                    c.witness_batch_position_major(d_c, d_l, out=out_c, chars_pm_stride=chars.shape[1])
                    inside.set()
                    freed.wait(60)
                    c.witness_batch_position_major(d_c, d_l, out=out_c, chars_pm_stride=chars.shape[1])
            g.replay()
            torch.cuda.synchronize()
            result["ok"] = True
        except Exception as e:      # noqa: BLE001
            result["error"] = repr(e)
            inside.set()

    with torch.cuda.stream(s1):       # (first launches allocate the contexts' scratch, and a context that changes streams waits for the old one on the host: not while a capture is open)
        a.witness_batch_position_major(d_c, d_l, out=out_a, chars_pm_stride=chars.shape[1])
    c.witness_batch_position_major(d_c, d_l, out=out_c, chars_pm_stride=chars.shape[1])
    torch.cuda.synchronize()
    t = threading.Thread(target=capturer)
    t.start()
    assert inside.wait(60) and "error" not in result, result
    import time
    t0 = time.perf_counter()
    with torch.cuda.stream(s1):
        for _ in range(100):          # ~8 ms of queued work writing out_a (a stream query is itself refused while the other thread captures: the clock tells)
            a.witness_batch_position_major(d_c, d_l, out=out_a, chars_pm_stride=chars.shape[1])
    t_queued = time.perf_counter() - t0
    del out_a                         # hrx_device_free of both sub-buffers while the other thread's capture is open (the collector's own pass below is not on the clock)
    t_freed = time.perf_counter() - t0
    gc.collect()
    assert t_freed < 0.006, "hrx_device_free waited (queued %.4f s, freed %.4f s)" % (t_queued, t_freed)
    freed.set()
    t.join(120)
    assert result.get("ok"), result
    orec, omsk, ost = OracleDefs.from_files(oracle, CFG_1).witness_batch(chars[:2048], lens[:2048], M, threads=os.cpu_count() or 8)
    assert np.array_equal(out_c[2].cpu().numpy().view(np.uint64)[:2048], ost)
    del keep


def test_placement_switches_per_context(hra):
    """hrx_ctx_set_placement: off / on and the budget of a walk per context, through the C ABI (not only HRX_PLACE=0 in the environment), and
    hrx_place_report.capped saying which bound ended a walk."""
    import gc
    import torch
    dev = torch.device("cuda", 0)
    M = 2048
    gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
    cfg = _cfg(hra, CFG_23, M)
    B = 131072                                                          # records 2 GiB: a direct walk (candidates = 512-MiB masked-row buffers)
    cfg.set_placement(walk=False)
    out = cfg.alloc_outputs_position_major(B, dev)
    rep = cfg.last_placement_report()
    assert rep["searched"] == 0 and rep["steps"] == 0 and rep["capped"] == 0           # two plain allocations, nothing measured
    del out; gc.collect()
    cfg.set_placement(walk=True, max_bytes=3 * (512 << 20) + 1)                         # room for three candidates at a time
    out = cfg.alloc_outputs_position_major(B, dev)
    rep = cfg.last_placement_report()
    assert rep["searched"] == 1 and 1 <= rep["steps"] <= 3 and rep["peak_candidate_bytes"] <= 3 * (512 << 20) + 1
    assert rep["steps"] < 3 or rep["accepted"] or (rep["capped"] & hra.PLACE_CAPPED_BYTES)
    del out; gc.collect()
    cfg.set_placement(walk=True, max_ms=0.001)                                          # the time bound ends the walk behind its first candidate
    out = cfg.alloc_outputs_position_major(B, dev)
    rep = cfg.last_placement_report()
    assert rep["searched"] == 1 and rep["steps"] == 1 and (rep["capped"] & hra.PLACE_CAPPED_TIME) and rep["best_us"] > 0
    del out; gc.collect()
    cfg.set_placement(walk=True)                                                        # defaults again: an uncapped walk reports no bound
    out = cfg.alloc_outputs_position_major(B, dev)
    rep = cfg.last_placement_report()
    assert rep["searched"] == 1 and rep["steps"] >= 1 and (rep["capped"] == 0 or not rep["accepted"] or rep["steps"] >= 8)
    import ctypes as C
    assert hra.lib.hrx_ctx_set_placement(cfg._ctx, 7, 0, C.c_double(0.0)) == hra.HRX_ERR_ARG
    assert hra.lib.hrx_ctx_set_placement(cfg._ctx, 1, 0, C.c_double(-1.0)) == hra.HRX_ERR_ARG
    assert hra.lib.hrx_ctx_set_placement(None, 1, 0, C.c_double(0.0)) == hra.HRX_ERR_ARG


def test_one_context_per_thread_on_one_device(hra, oracle):
    """How a multi-threaded prover uses the library (a context serves one stream at a time; halo2 synthesizes from several threads): four threads, each with a context
    and a stream of its own, allocate bench-sized outputs (the device's shared arena pair: pool mutex, arena mutex), launch, free and allocate again, concurrently —
    ctypes drops the GIL inside every call.  Every thread's rows against the oracle."""
    import gc
    import threading
    import torch
    from halo2_regex_amd import synth
    dev = torch.device("cuda", 0)
    M, B, T = 1024, 65536, 4
    chars, lens = synth.reveal_stress(B, M - 1, seed=21)
    orec, omsk, ost = OracleDefs.from_files(oracle, CFG_1).witness_batch(chars, lens, M, threads=os.cpu_count() or 8)
    d_c = hra.chars_to_position_major(torch.from_numpy(chars).to(dev))
    d_l = torch.from_numpy(lens.astype(np.int32)).to(dev)
    d_orec, d_omsk = torch.from_numpy(orec.view(np.int32)).to(dev), torch.from_numpy(omsk.view(np.int16)).to(dev)
    ok = torch.from_numpy((ost & np.uint64(0xff)) == 0).to(dev)
    torch.cuda.synchronize()
    errors, reports = [], []
    start = threading.Barrier(T)

    def worker(k):
        try:
            torch.cuda.set_device(0)
            cfg = _cfg(hra, CFG_1, M)
            stream = torch.cuda.Stream(device=dev)
            start.wait()
            for it in range(3):
                # everything of this thread on ITS stream — torch tensors included: the status tensor comes from torch's caching allocator, which hands a block freed on
                # one stream to the next request on the same stream without waiting; allocated on the shared default stream and written on a side stream it was clobbered,
                # now and then, by another thread's still-pending temporaries (seen: 65535 status words zero, one run in eight)
                with torch.cuda.stream(stream):
                    out = cfg.alloc_outputs_position_major(B, dev)
                    reports.append(cfg.last_placement_report()["searched"])
                    out[0].fill_(-1); out[1].fill_(-1); out[2].fill_(-1)     # (a reused range still holds an earlier iteration's — correct — rows: the launch must write them again)
                    rec, msk, st = cfg.witness_batch_position_major(d_c, d_l, out=out, chars_pm_stride=chars.shape[1], stream=stream)
                    r2, m2 = hra.position_major_to_string_major(rec, msk, B, M, 1)
                    got = st.cpu().numpy().view(np.uint64)
                    rows_ok = bool(torch.equal(r2[ok], d_orec[ok]) and torch.equal(m2[ok], d_omsk[ok]))
                stream.synchronize()
                if not np.array_equal(got, ost):
                    bad = np.nonzero(got != ost)[0]
                    errors.append("thread %d iteration %d: %d status words differ, first at string %d: 0x%x for 0x%x" % (k, it, bad.size, bad[0], int(got[bad[0]]), int(ost[bad[0]])))
                if not rows_ok:
                    errors.append("thread %d iteration %d: rows differ" % (k, it))
                del out, rec, msk, st, r2, m2      # hrx_device_free from this thread while the others allocate
            del cfg
        except Exception as e:   # noqa: BLE001 — reported by the main thread
            errors.append("thread %d: %r" % (k, e))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(T)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not any(t.is_alive() for t in threads), "a worker hangs"
    assert not errors, errors
    assert sorted(set(reports)) in ([1, 2], [2]) and reports.count(1) <= 1      # at most ONE walk for the device (none if an earlier context's pair is still there): freed sub-buffers go back into the pair (hrx_arena_alloc.hpp), twelve allocations never fill it
    gc.collect()


def test_multi_device_driver_device_resident_shards(hra, oracle):
    """hrx_multi_witness_batch_device: device pointers per shard, one stream per shard, no PCIe copy, no collective — three shards
    on the one device (two position-major blocks' worth of strings in the middle shard), every string against the oracle."""
    import torch
    from halo2_regex_amd import synth
    dev = torch.device("cuda", 0)
    M = 136
    chars, lens = synth.reveal_stress(3000, 130, seed=13)
    cfg = _cfg(hra, CFG_A, M)
    multi = hra.MultiDevice(cfg, [0, 0, 0])
    orec, omsk, ost = OracleDefs.from_files(oracle, CFG_A).witness_batch(chars, lens, M, threads=8)
    ok = (ost & np.uint64(0xff)) == 0
    for layout_pm in (True, False):
        shards, cuts = [], []
        for r in range(3):
            b, c = hra.shard_range(len(lens), 3, r)
            cuts.append((b, c))
            d_c = torch.from_numpy(chars[b:b + c]).to(dev)
            d_l = torch.from_numpy(lens[b:b + c].astype(np.int32)).to(dev)
            if layout_pm:
                shards.append((hra.chars_to_position_major(d_c), d_l, cfg.alloc_outputs_position_major(c, dev)))
            else:
                shards.append((d_c, d_l, cfg.alloc_outputs(c, dev)))
        if layout_pm:
            # chars_stride omitted: it follows from the buffer sizes (never a silent fallback); shards that disagree are refused
            bad = [(shards[0][0][:-16], shards[0][1], shards[0][2])] + shards[1:]
            with pytest.raises(hra.HrxError):
                multi.witness_batch_device(bad)
            multi.witness_batch_device(shards)
        else:
            multi.witness_batch_device(shards, layout=hra.LAYOUT_STRING_MAJOR)
        multi.synchronize()
        for (b, c), (_, _, (rec, msk, st)) in zip(cuts, shards):
            if layout_pm:
                rec, msk = hra.position_major_to_string_major(rec, msk, c, M, 2)
            assert np.array_equal(st.cpu().numpy().view(np.uint64), ost[b:b + c])
            k = ok[b:b + c]
            assert np.array_equal(rec.cpu().numpy().view(np.uint32)[k], orec[b:b + c][k])
            assert np.array_equal(msk.cpu().numpy().view(np.uint16)[k], omsk[b:b + c][k])


def test_bench_runs_bare_with_two_ranks_on_one_device():
    """`python3 bench.py --gpus 2` as a bare command: the parent touches no GPU and spawns one child per rank (here both ranks
    mapped onto device 0: HRX_BENCH_DEVICES=0,0), gloo carries the barrier, the data path has no collective; the line
    aggregates both ranks and rank 0's timed buffers are verified against the oracle."""
    import json
    import subprocess
    import sys
    from oracle_lib import ROOT
    env = dict(os.environ, HRX_BENCH_DEVICES="0,0")
    env.pop("HRX_DEBUG_FLAGS", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--batch", "16384",
                          "--no-cpu-baseline", "--no-spread"], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and len(line["per_rank"]) == 2
    assert sum(r["rows"] for r in line["per_rank"]) == 2 * 10 * 16384 * 1023
    assert abs(line["value"] - sum(r["rows"] for r in line["per_rank"]) / max(r["elapsed_s"] for r in line["per_rank"])) < 1e-6 * line["value"]
    assert line["verified"]["bit_exact"] is True and line["debug_flags"] is None


def test_bench_strong_scaling_shards_one_job_over_the_ranks():
    """`python3 bench.py --gpus 2 --scaling strong --config headers3`: --batch strings IN TOTAL, sharded by string index over the ranks
    (hrx_shard_range; BASELINE configs[3] is such a job: 262144 strings over 8 GPUs), every rank verifies its own shard against the
    oracle; both ranks on device 0 here.  With one rank the strong and the weak line are the same job."""
    import json
    import subprocess
    import sys
    from oracle_lib import ROOT
    env = dict(os.environ, HRX_BENCH_DEVICES="0,0")
    env.pop("HRX_DEBUG_FLAGS", None)
    common = ["--steps", "6", "--warmup", "2", "--config", "headers3", "--batch", "6000", "--len", "2000", "--rows", "2048", "--sets", "2",
              "--no-cpu-baseline", "--no-spread", "--no-pmc"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scaling", "strong", "--verify-all-ranks"] + common,
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and [r["strings"] for r in line["per_rank"]] == [3000, 3000]
    assert [r["shard_begin"] for r in line["per_rank"]] == [0, 3000] and line["per_rank"][1]["verified"] is True and line["verified"]["bit_exact"] is True
    assert sum(r["rows"] for r in line["per_rank"]) == 6 * 6000 * 2000
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--scaling", "strong"] + common, capture_output=True, text=True, env=env, timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    l1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert l1["per_rank"][0]["strings"] == 6000 and l1["per_rank"][0]["rows"] == 6 * 6000 * 2000 and l1["verified"]["bit_exact"] is True


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
    # (1) every string, bit-exact against the oracle
    o = OracleDefs.from_files(oracle, CFG_1)
    orec, omsk, ost = o.witness_batch(chars, lens, M, threads=os.cpu_count() or 1)
    assert np.array_equal(orec, rec_h) and np.array_equal(omsk, msk_h) and np.array_equal(ost, st_h)
    del orec, omsk
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


# ---------------------------------------------------------------------------------------------
# The shapes profiles/*config_sweep* quotes throughput at, at FULL size: every string against the oracle and through the
# integer MockProver.  (Inputs: one generated block, rotated per block — tests/_rolled_blocks — so that generating 2 GiB of
# planted text does not dominate the test; every block still has its own ragged strings.)
# ---------------------------------------------------------------------------------------------
def test_full_size_cfg3_two_defs_2pow20_strings_16_blocks(hra, oracle):
    """BASELINE configs[2]: regex2 + regex3 with substr extraction, 2^20 x 2048-byte strings = 16 blocks of the position-major
    buffers (2 GiB in, 21 GiB out), D = 2, one launch."""
    from halo2_regex_amd import synth
    n, M = 2047, 2048
    base_c, base_l = synth.regex23_planted(hra.PM_BLOCK, n, seed=1, stride=2048)
    blocks = _rolled_blocks(base_c, base_l, 16, seed=3)
    cfg = _cfg(hra, CFG_23, M)
    assert "witness_pm_kernel<2, false, true, false, false, false>" in cfg.describe_launch(1 << 20, layout=3)
    st = _full_check(hra, OracleDefs.from_files(oracle, CFG_23), cfg, blocks, M, 2, need_accept=0.0)
    assert len(st) == 1 << 20 and (st & np.uint64(0xff) == 0).all()


def test_full_size_cfg4_share_32768_strings_of_32768_bytes(hra, oracle):
    """BASELINE configs[3], one GPU's share: D = 3 header definitions (5 substrs), 32768 strings x 32768 bytes (1 GiB in, 15 GiB out)."""
    from halo2_regex_amd import synth
    from test_substr_gen import header_def_texts
    texts = header_def_texts()
    defs = [hra.RegexDefs(hra.AllstrRegexDef(a), [hra.SubstrRegexDef(t) for t in subs]) for a, subs in texts]
    n, M, B = 32767, 32768, 32768
    base_c, base_l = synth.headers_planted(4096, n, seed=3, stride=M)
    parts = _rolled_blocks(base_c, base_l, B // 4096, seed=4)
    chars = np.concatenate([c for c, _ in parts])
    lens = np.concatenate([l for _, l in parts])
    del parts
    cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
    st = _full_check(hra, OracleDefs(oracle, texts), cfg, [(chars, lens)], M, 3, need_accept=0.9)
    assert (st & np.uint64(0xff) == 0).all()


def test_full_size_cfg5_two_substr_defs_131072_strings_of_4096_bytes(hra, oracle):
    """BASELINE configs[4] with TWO substring definitions (SURVEY §8d cfg 5: "1-2 substr defs drawn as random transition subsets"): 100 tagged pairs each — ids 1 and 2 in
    one def, so the BYTE kernel's finisher holds two tiles of id bytes instead of three of bits (no byte_one_id) — at full size, every string, both layouts' kernels."""
    from halo2_regex_amd import synth
    allb = np.arange(256, dtype=np.uint8)
    allstr, subs = synth.random_dfa_multi(256, seed=2, alphabet=allb, n_substr_pairs=100, n_substrs=2)
    defs = [hra.RegexDefs(hra.AllstrRegexDef(allstr), [hra.SubstrRegexDef(t) for t in subs])]
    n, M = 4095, 4096
    base_c, base_l = synth.noise(hra.PM_BLOCK, n, seed=4, alphabet=allb, stride=4096)
    cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
    assert "witness_pm_kernel<1, false, false, false, false, true>" in cfg.describe_launch(131072, layout=3)     # the BYTE table (200 tagged pairs in all)
    o = OracleDefs(oracle, [(allstr, subs)])
    st = _full_check(hra, o, cfg, _rolled_blocks(base_c, base_l, 2, seed=9), M, 1)
    assert (st & np.uint64(0xff) == 0).all()
    # both ids turn up in the masked rows of the batch (the oracle's view of the first strings)
    orec, omsk, _ = o.witness_batch(base_c[:256], base_l[:256], M, threads=os.cpu_count() or 8)
    assert set(np.unique(orec[..., 0] >> 16 & 0xff)) >= {0, 1, 2} and set(np.unique(omsk >> 8)) >= {1, 2}
    # string-major: the walker/storer kernel on the BYTE table
    assert "witness_split_kernel<1, 32, true>" in cfg.describe_launch(8192, layout=0)
    _full_check(hra, o, cfg, [(base_c[:8192], base_l[:8192])], M, 1, position_major=False)


def test_full_size_cfg5_dfa256_131072_strings_of_4096_bytes(hra, oracle):
    """BASELINE configs[4] shape: 256-state x 256-symbol DFA (HALF table in LDS), 131072 x 4096-byte strings = 2 blocks; the random
    substring definition makes the optimistic end-mask protocol repair rows every few tiles."""
    from halo2_regex_amd import synth
    allb = np.arange(256, dtype=np.uint8)
    allstr, sub = synth.random_dfa(256, seed=2, alphabet=allb, n_substr_pairs=200)
    defs = [hra.RegexDefs(hra.AllstrRegexDef(allstr), [hra.SubstrRegexDef(sub)])]
    n, M = 4095, 4096
    base_c, base_l = synth.noise(hra.PM_BLOCK, n, seed=2, alphabet=allb, stride=4096)
    cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
    assert "witness_pm_kernel<1, false, false, false, false, true>" in cfg.describe_launch(131072, layout=3)     # the BYTE table: walker + loader + finisher
    st = _full_check(hra, OracleDefs(oracle, [(allstr, [sub])]), cfg, _rolled_blocks(base_c, base_l, 2, seed=5), M, 1)
    assert (st & np.uint64(0xff) == 0).all()
