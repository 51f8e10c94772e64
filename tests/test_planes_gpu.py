"""RECORD PLANES (include/hrx.h hrx_witness_batch_device_planes): every def's records in a buffer of its own — the values of src/lib.rs:387-519's per-def columns,
unchanged; only where they lie differs.  Every kernel that writes position-major records is driven through the same batches with both layouts and the rows are compared
with the oracle's (bit-exact).  Also: the def-parallel kernel with a combiner wave of its own (HRX_OPT_PMD_COMBINER_WAVE) on two and three defs."""
import os

import numpy as np
import pytest

from oracle_lib import OracleDefs, DFA_DIR
from test_parity_gpu import CFG_A, CFG_23, CFG_123, CFG_D4, CFG_D5, CFG_D8, HDR, NO_HOST, _cfg, hra  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _device_paths_only(monkeypatch):
    monkeypatch.setenv("HRX_DEBUG_FLAGS", str(NO_HOST))


def _flags(bits=0):
    return str(int(bits) | NO_HOST)


def _planes_rows(hra, cfg, chars, lens, M, pm_input, out=None):
    import torch
    dev = torch.device("cuda", 0)
    B = len(lens)
    if out is None and cfg.num_defs == 1:
        out = cfg.alloc_output_planes(B, dev, stripes=2)     # one def: its two row stripes
    stride = (chars.shape[1] + 15) // 16 * 16
    wide = torch.zeros((B, max(stride, 16)), dtype=torch.uint8, device=dev)
    wide[:, :chars.shape[1]] = torch.from_numpy(chars).to(dev)
    d_lens = torch.from_numpy(lens.astype(np.int32)).to(dev)
    if pm_input:
        planes, msk, st = cfg.witness_batch_planes(hra.chars_to_position_major(wide), d_lens, chars_pm_stride=wide.shape[1], out=out)
    else:
        planes, msk, st = cfg.witness_batch_planes(wide, d_lens, out=out)
    torch.cuda.synchronize()
    rec, m = hra.planes_to_string_major(planes, msk, B, M, D=cfg.num_defs)
    return rec.cpu().numpy().view(np.uint32), m.cpu().numpy().view(np.uint16), st.cpu().numpy().view(np.uint64), planes, msk


def _check_planes(hra, oracle, names, chars, lens, M, cfg=None):
    cfg = cfg or _cfg(hra, names, M)
    o = OracleDefs.from_files(oracle, names)
    orec, omsk, ost = o.witness_batch(chars, lens, M, threads=os.cpu_count() or 1)
    ok = (ost & np.uint64(0xff)) == 0
    for pm_input in (False, True):
        rec, msk, st, planes, d_msk = _planes_rows(hra, cfg, chars, lens, M, pm_input)
        assert np.array_equal(st, ost)
        assert np.array_equal(rec[ok], orec[ok]) and np.array_equal(msk[ok], omsk[ok])
    # the host gather out of planes copied to the host as they are
    hp = [p.cpu().numpy() for p in planes]
    hm = d_msk.cpu().numpy()
    for b in list(range(0, len(lens), max(1, len(lens) // 17)))[:20]:
        if ok[b]:
            r1, m1 = hra.rows_of_string_planes(hp, hm, len(lens), M, b, D=cfg.num_defs)
            assert np.array_equal(r1, orec[b]) and np.array_equal(m1, omsk[b])


def _stress(synth, M, seed):
    chars, lens = synth.reveal_stress(500, min(M - 8, 320), seed=seed)
    h_c, h_l = synth.headers_planted(200, chars.shape[1] - 3, seed=5, stride=chars.shape[1])
    chars, lens = np.concatenate([chars, h_c]), np.concatenate([lens, h_l])
    chars[7, 50] = 250                                       # a byte no DFA has a transition for
    chars[9, 0] = 200
    lens[11] = M + 72                                        # n > M
    lens[12], lens[13] = 0, min(M, chars.shape[1])
    return chars, lens


@pytest.mark.parametrize("flags", [0, 0x4000000, 0x2000000, 0x2000000 | 0x1000, 0x2000000 | 0x80000], ids=["planner", "def-parallel", "regular", "dynamic-groups", "narrow-table"])
@pytest.mark.parametrize("names", [CFG_A, CFG_23, CFG_123, HDR], ids=["r1r2", "r2r3", "r1r2r3", "headers3"])
def test_record_planes_of_two_and_three_defs(hra, oracle, names, flags, monkeypatch):
    from halo2_regex_amd import synth
    monkeypatch.setenv("HRX_DEBUG_FLAGS", _flags(flags))
    for M in (328, 203, 64):
        chars, lens = _stress(synth, max(M, 72), seed=37 + M)
        _check_planes(hra, oracle, names, chars, lens, M)


@pytest.mark.parametrize("fin", [1, 2], ids=["combiner-wave", "last-walker-combines"])
@pytest.mark.parametrize("names", [CFG_A, CFG_123, HDR], ids=["r1r2", "r1r2r3", "headers3"])
def test_def_parallel_kernel_with_and_without_a_combiner_wave(hra, oracle, names, fin, monkeypatch):
    """HRX_OPT_PMD_COMBINER_WAVE: both variants of the def-parallel kernel on the WIDE table — one group per workgroup (a batch of at most one group per CU) and two —
    interleaved records and record planes, every string against the oracle; multi-round launches (groups beyond the grid) through the forced kernel."""
    from halo2_regex_amd import synth
    D = len(names)
    for force, B_rep in ((0, 1), (0x4000000, 1), (0x4000000, 60)):
        monkeypatch.setenv("HRX_DEBUG_FLAGS", _flags(force))
        for M in (328, 203):
            cfg = _cfg(hra, names, M)
            cfg.set_option(hra.OPT_PMD_COMBINER_WAVE, fin)
            assert cfg.get_option(hra.OPT_PMD_COMBINER_WAVE) == fin
            chars, lens = _stress(synth, M, seed=11 + M)
            chars, lens = np.tile(chars, (B_rep, 1)), np.tile(lens, B_rep)
            d = cfg.describe_launch(len(lens), layout=3)
            assert d.startswith("hrx::witness_pmd_kernel<%d, false, %s, false>" % (D, "true" if fin == 1 else "false")), d
            cfg0 = _cfg(hra, names, M)      # the default: a combiner wave for record planes, none for the interleaved records
            assert cfg0.describe_launch(len(lens), layout=3 | hra.LAYOUT_RECORD_PLANES).startswith("hrx::witness_pmd_kernel<%d, false, true, false>" % D)
            assert cfg0.describe_launch(len(lens), layout=3).startswith("hrx::witness_pmd_kernel<%d, false, false, false>" % D)
            _check_planes(hra, oracle, names, chars, lens, M, cfg=cfg)
            # ... and the interleaved layout through the same context
            import torch
            dev = torch.device("cuda", 0)
            o = OracleDefs.from_files(oracle, names)
            orec, omsk, ost = o.witness_batch(chars, lens, M, threads=os.cpu_count() or 1)
            wide = torch.zeros((len(lens), (chars.shape[1] + 15) // 16 * 16), dtype=torch.uint8, device=dev)
            wide[:, :chars.shape[1]] = torch.from_numpy(chars).to(dev)
            rec, msk, st = cfg.witness_batch_position_major(wide, torch.from_numpy(lens.astype(np.int32)).to(dev))
            torch.cuda.synchronize()
            g_r, g_m = hra.position_major_to_string_major(rec, msk, len(lens), M, D)
            ok = (ost & np.uint64(0xff)) == 0
            assert np.array_equal(st.cpu().numpy().view(np.uint64), ost)
            assert np.array_equal(g_r.cpu().numpy().view(np.uint32)[ok], orec[ok]) and np.array_equal(g_m.cpu().numpy().view(np.uint16)[ok], omsk[ok])


@pytest.mark.parametrize("names", [CFG_D4, CFG_D5, CFG_D8], ids=["D4", "D5", "D8"])
def test_record_planes_of_four_to_eight_defs(hra, oracle, names):
    from halo2_regex_amd import synth
    for M in (328, 203):
        chars, lens = _stress(synth, M, seed=5 + M)
        _check_planes(hra, oracle, names, chars, lens, M)


def test_record_planes_over_two_blocks_of_strings(hra, oracle):
    """70000 strings: two blocks of the position-major buffers, every plane blocked like a one-def records buffer; multi-round launches."""
    from halo2_regex_amd import synth
    M, B = 136, 70000
    chars, lens = synth.ragged(B, M, seed=29)
    for names in (CFG_A, CFG_123, CFG_D4):
        _check_planes(hra, oracle, names, chars, lens, M)


def test_record_planes_argument_checks_and_one_def(hra, oracle):
    import torch
    from halo2_regex_amd import synth
    from test_parity_gpu import CFG_1, CFG_D13
    dev = torch.device("cuda", 0)
    M = 200
    chars, lens = synth.ragged(300, M, seed=3)
    _check_planes(hra, oracle, CFG_1, chars, lens, M)          # one def: two row stripes
    cfg1 = _cfg(hra, CFG_1, M)
    one = cfg1.alloc_output_planes(300, dev)                    # ... or one buffer: the position-major records buffer itself
    assert len(one[0]) == 1 and one[0][0].numel() == (M + 3) // 4 * 300 * 4
    d_l = torch.from_numpy(lens.astype(np.int32)).to(dev)
    w = torch.zeros((300, 208), dtype=torch.uint8, device=dev); w[:, :chars.shape[1]] = torch.from_numpy(chars).to(dev)
    p1, m1, s1 = cfg1.witness_batch_planes(w, d_l, out=one)
    r0, m0, s0 = cfg1.witness_batch_position_major(w, d_l)
    torch.cuda.synchronize()
    assert torch.equal(p1[0], r0) and torch.equal(m1, m0) and torch.equal(s1, s0)
    cfg = _cfg(hra, CFG_A, M)
    planes, msk, st = cfg.alloc_output_planes(300, dev)
    assert len(planes) == 2 and planes[0].numel() == (M + 3) // 4 * 300 * 4
    wide = torch.zeros((300, 208), dtype=torch.uint8, device=dev)
    d_lens = torch.zeros(300, dtype=torch.int32, device=dev)
    with pytest.raises(hra.HrxError):       # one plane too few
        cfg.witness_batch_planes(wide, d_lens, out=(planes[:1], msk, st))
    cfg13 = _cfg(hra, CFG_D13, M)           # more than eight defs: passes over groups — no record planes
    with pytest.raises(hra.HrxError):
        cfg13.witness_batch_planes(wide, d_lens, out=([torch.empty_like(planes[0]) for _ in range(13)], msk, st))


def test_placed_record_planes_at_a_multi_gigabyte_size(hra, oracle):
    """hrx_alloc_output_planes walks (planes of 1 GiB): three planes + masked rows, each measured against the ones kept before it; the launch into them writes the oracle's rows."""
    import torch
    from halo2_regex_amd import synth
    dev = torch.device("cuda", 0)
    M, B = 4096, 65536
    base_c, base_l = synth.headers_planted(2048, M - 1, seed=3, stride=M)
    chars, lens = np.tile(base_c, (B // 2048, 1)), np.tile(base_l, B // 2048)
    cfg = _cfg(hra, HDR, M)
    out = cfg.alloc_output_planes(B, dev)
    rep = cfg.last_placement_report()
    assert rep["searched"] == 1 and rep["steps"] >= 3 and rep["ref_gbs"] > 1000 and rep["best_gbs"] > 1000
    assert len({p.data_ptr() for p in out[0]}) == 3
    d_chars = hra.chars_to_position_major(torch.from_numpy(chars).to(dev))
    d_lens = torch.from_numpy(lens.astype(np.int32)).to(dev)
    planes, msk, st = cfg.witness_batch_planes(d_chars, d_lens, out=out, chars_pm_stride=M)
    torch.cuda.synchronize()
    o = OracleDefs.from_files(oracle, HDR)
    orec, omsk, ost = o.witness_batch(base_c, base_l, M, threads=os.cpu_count() or 1)
    rec, m = hra.planes_to_string_major(planes, msk, B, M, D=3)
    t_rec, t_msk = torch.from_numpy(orec.view(np.int32)).to(dev), torch.from_numpy(omsk.view(np.int16)).to(dev)
    assert (ost & np.uint64(0xff) == 0).all()
    for k in range(0, B, 2048):
        assert torch.equal(rec[k:k + 2048], t_rec) and torch.equal(m[k:k + 2048], t_msk)
    assert (st.cpu().numpy().view(np.uint64) == np.tile(ost, B // 2048)).all()
    # the no-compute pass over the planes (dealt like the def-parallel kernel for this shape): runs, and overwrites what it is pointed at
    cfg.traffic_pass_planes(d_chars, B, out, M)
    torch.cuda.synchronize()
    rec2, _ = hra.planes_to_string_major(planes, msk, B, M, D=3)
    assert not torch.equal(rec2[:2048], t_rec)
    cfg.set_placement(walk=False)
    out2 = cfg.alloc_output_planes(B, dev)
    assert cfg.last_placement_report()["searched"] == 0 and len(out2[0]) == 3


@pytest.mark.parametrize("names", ["HDR", "ONE"])
def test_buffers_chosen_with_the_callers_batch(hra, oracle, names):
    """hrx_alloc_output_planes_for_batch: the choosing launches run the caller's batch (which they only read); three defs and — through the same pool — one def's single buffer; the
    launch into the kept buffers writes the oracle's rows.  Wrong arguments are refused before anything is allocated."""
    import ctypes as C
    import torch
    from halo2_regex_amd import synth
    from test_parity_gpu import CFG_1
    dev = torch.device("cuda", 0)
    names = HDR if names == "HDR" else CFG_1
    D = len(names)
    M, B = (4096, 65536) if D == 3 else (8192, 65536)
    base_c, base_l = synth.headers_planted(1024, M - 1, seed=5, stride=M)
    chars, lens = np.tile(base_c, (B // 1024, 1)), np.tile(base_l, B // 1024)
    cfg = _cfg(hra, names, M)
    d_chars = hra.chars_to_position_major(torch.from_numpy(chars).to(dev))
    d_lens = torch.from_numpy(lens.astype(np.int32)).to(dev)
    before = d_chars.clone()
    out = cfg.alloc_output_planes(B, dev, chars=d_chars, lens=d_lens, chars_pm_stride=M)
    rep = cfg.last_placement_report()
    assert rep["searched"] == 1 and rep["steps"] >= 3 and len({p.data_ptr() for p in out[0]}) == D
    assert torch.equal(before, d_chars)
    planes, msk, st = cfg.witness_batch_planes(d_chars, d_lens, out=out, chars_pm_stride=M)
    torch.cuda.synchronize()
    o = OracleDefs.from_files(oracle, names)
    orec, omsk, ost = o.witness_batch(base_c, base_l, M, threads=os.cpu_count() or 1)
    rec, m = hra.planes_to_string_major(planes, msk, B, M, D=D)
    t_rec, t_msk = torch.from_numpy(orec.view(np.int32)).to(dev), torch.from_numpy(omsk.view(np.int16)).to(dev)
    for k in range(0, B, 1024):
        assert torch.equal(rec[k:k + 1024], t_rec) and torch.equal(m[k:k + 1024], t_msk)
    assert (st.cpu().numpy().view(np.uint64) == np.tile(ost, B // 1024)).all()
    arr, pmk = (C.c_void_p * D)(), C.c_void_p()
    lib = hra.lib
    assert lib.hrx_alloc_output_planes_for_batch(cfg._ctx, 3, None, M, d_lens.data_ptr(), B, M, D, arr, C.byref(pmk)) == hra.HRX_ERR_ARG
    assert lib.hrx_alloc_output_planes_for_batch(cfg._ctx, 2, d_chars.data_ptr(), M, d_lens.data_ptr(), B, M, D, arr, C.byref(pmk)) == hra.HRX_ERR_ARG
    assert lib.hrx_alloc_output_planes_for_batch(cfg._ctx, 3, d_chars.data_ptr(), M + 8, d_lens.data_ptr(), B, M, D, arr, C.byref(pmk)) == hra.HRX_ERR_ARG
    assert lib.hrx_alloc_output_planes_for_batch(cfg._ctx, 3, d_chars.data_ptr(), M, d_lens.data_ptr(), B, M, D + 2, arr, C.byref(pmk)) == hra.HRX_ERR_ARG
    # a batch too small to be worth a walk: plain allocations, the batch is not launched (256 strings of it)
    assert lib.hrx_alloc_output_planes_for_batch(cfg._ctx, 3, d_chars.data_ptr(), M, d_lens.data_ptr(), 256, M, D, arr, C.byref(pmk)) == 0
    assert cfg.last_placement_report()["searched"] == 0 and all(arr[d] for d in range(D)) and pmk.value
    for d in range(D):
        lib.hrx_device_free(arr[d])
    lib.hrx_device_free(pmk)


@pytest.mark.parametrize("flags", [0, 0x200000, 0x400000, 0x40000, 0x1000, 0x40000000, 0x2000, 0x8000], ids=["planner", "wide", "half", "global-table", "dynamic-groups", "pair-step-asked", "byte", "no-byte"])
def test_row_stripes_of_one_def_on_every_table_format(hra, oracle, flags, monkeypatch):
    """One def in TWO ROW STRIPES (quad q of a string in buffer q % 2 at slot q / 2): every table format of the loader / walker / finisher kernel, odd and even numbers of quads, ragged
    and failing strings, both input layouts, two blocks of strings; the pair-step kernel does not write stripes (the planner must not pick it even when asked to)."""
    from halo2_regex_amd import synth
    from test_parity_gpu import CFG_1, CFG_3
    monkeypatch.setenv("HRX_DEBUG_FLAGS", _flags(flags))
    for names in (CFG_1, CFG_3):
        for M in (328, 203, 64, 4, 5):
            chars, lens = _stress(synth, max(M, 72), seed=41 + M)
            if M < 72:
                lens = np.minimum(lens, M + 3)
            cfg = _cfg(hra, names, M)
            assert "witness_pp_kernel" not in cfg.describe_launch(len(lens), layout=3 | hra.LAYOUT_RECORD_PLANES)
            _check_planes(hra, oracle, names, chars, lens, M, cfg=cfg)
    M, B = 136, 70000
    chars, lens = synth.ragged(B, M, seed=31)
    _check_planes(hra, oracle, CFG_1, chars, lens, M)


def test_row_stripes_of_a_256_state_dfa(hra, oracle):
    """cfg 5's table format (BYTE: a 1-byte next-state table + perfect-hash pair tags) with the records in two row stripes, 4096-row strings."""
    import torch
    from halo2_regex_amd import synth
    dev = torch.device("cuda", 0)
    allb = np.arange(256, dtype=np.uint8)
    a_txt, sub_txt = synth.random_dfa(256, seed=2, alphabet=allb, n_substr_pairs=200)
    defs = [hra.RegexDefs(hra.AllstrRegexDef(a_txt.encode()), [hra.SubstrRegexDef(sub_txt.encode())])]
    M, B = 4096, 4096
    cfg = hra.RegexVerifyConfig.configure(M, defs, device=0)
    assert "true>" in cfg.describe_launch(B, layout=3 | hra.LAYOUT_RECORD_PLANES).split(" grid=")[0]      # the BYTE kernel
    chars, lens = synth.noise(B, M - 1, seed=4, alphabet=allb, stride=M)
    lens[:64] = np.random.default_rng(1).integers(0, M, 64)
    o = OracleDefs(oracle, [(a_txt.encode(), [sub_txt.encode()])])
    orec, omsk, ost = o.witness_batch(chars, lens, M, threads=os.cpu_count() or 1)
    d_chars = hra.chars_to_position_major(torch.from_numpy(chars).to(dev))
    planes, msk, st = cfg.witness_batch_planes(d_chars, torch.from_numpy(lens.astype(np.int32)).to(dev), chars_pm_stride=M, out=cfg.alloc_output_planes(B, dev, stripes=2))
    torch.cuda.synchronize()
    assert len(planes) == 2
    rec, m = hra.planes_to_string_major(planes, msk, B, M, D=1)
    assert np.array_equal(st.cpu().numpy().view(np.uint64), ost)
    assert np.array_equal(rec.cpu().numpy().view(np.uint32), orec) and np.array_equal(m.cpu().numpy().view(np.uint16), omsk)


def test_small_row_stripes_come_out_of_measured_arenas(hra, oracle):
    """The bench line's sizes (stripes of 128 MiB): carved out of the device's stripe arenas — three 2-GiB blocks chosen once by their pairings; later requests are served from them
    (searched = 2), hrx_device_free gives the ranges back."""
    import torch
    dev = torch.device("cuda", 0)
    from test_parity_gpu import CFG_1
    M, B = 1024, 65536
    cfg = _cfg(hra, CFG_1, M)
    outs = [cfg.alloc_output_planes(B, dev, stripes=2) for _ in range(3)]
    reps = []
    for _ in range(2):
        outs.append(cfg.alloc_output_planes(B, dev, stripes=2))
        reps.append(cfg.last_placement_report())
    assert all(len(o[0]) == 2 for o in outs) and reps[-1]["searched"] == 2
    ptrs = sorted(p.data_ptr() for o in outs for p in o[0] + [o[1]])
    assert len(set(ptrs)) == len(ptrs)
    first = outs[0][0][0].data_ptr()
    del outs[0]
    torch.cuda.synchronize()
    again = cfg.alloc_output_planes(B, dev, stripes=2)
    assert first in (again[0][0].data_ptr(), again[0][1].data_ptr(), again[1].data_ptr()) or cfg.last_placement_report()["searched"] == 2


@pytest.mark.parametrize("names", [CFG_A, CFG_123, HDR], ids=["r1r2", "r1r2r3", "headers3"])
def test_record_planes_through_the_chunked_launch(hra, oracle, names, monkeypatch):
    """Few long strings are cut into chunks (scout -> compose -> walk over the chunks -> stitch -> repair: csrc/hrx_kernel_spec.hip); the walk writes record planes like any other
    launch and the repair reads the finished records out of the planes.  Forced 4-tile chunks over reveal-stress strings (revealed parts straddling chunk borders: the repair runs), and
    the planner's own choice for a batch of few long strings."""
    from halo2_regex_amd import synth
    monkeypatch.setenv("HRX_DEBUG_FLAGS", _flags(0x80))
    for M in (1024, 2048):
        chars, lens = synth.reveal_stress(300, M - 8, seed=7 + M)
        h_c, h_l = synth.headers_planted(100, chars.shape[1] - 3, seed=9, stride=chars.shape[1])
        chars, lens = np.concatenate([chars, h_c]), np.concatenate([lens, h_l])
        chars[7, 700] = 250
        lens[12], lens[13] = 0, min(M, chars.shape[1])
        cfg = _cfg(hra, names, M)
        assert "chunked=" in cfg.describe_launch(len(lens), layout=3 | hra.LAYOUT_RECORD_PLANES)
        _check_planes(hra, oracle, names, chars, lens, M, cfg=cfg)
    monkeypatch.setenv("HRX_DEBUG_FLAGS", _flags())
    M = 8192
    chars, lens = synth.headers_planted(512, M - 1, seed=11, stride=M)
    cfg = _cfg(hra, names, M)
    assert "chunked=" in cfg.describe_launch(512, layout=3 | hra.LAYOUT_RECORD_PLANES)
    _check_planes(hra, oracle, names, chars, lens, M, cfg=cfg)


@pytest.mark.parametrize("names,stripes", [(CFG_A, None), (CFG_123, None), (None, 2)], ids=["D2-planes", "D3-planes", "D1-stripes"])
def test_field_cells_out_of_record_planes(hra, oracle, names, stripes):
    """hrx_fr_columns_device_planes (SURVEY §8 f4 for the planes layout): the cells equal the ones hrx_fr_columns_device makes out of the interleaved buffers of the same batch."""
    import torch
    from halo2_regex_amd import synth
    from test_parity_gpu import CFG_1
    names = names or CFG_1
    dev = torch.device("cuda", 0)
    M, B = 200, 700
    chars, lens = synth.reveal_stress(B, M - 8, seed=21)
    cfg = _cfg(hra, names, M)
    stride = (chars.shape[1] + 15) // 16 * 16
    wide = torch.zeros((B, stride), dtype=torch.uint8, device=dev)
    wide[:, :chars.shape[1]] = torch.from_numpy(chars).to(dev)
    d_lens = torch.from_numpy(lens.astype(np.int32)).to(dev)
    c_pm = hra.chars_to_position_major(wide)
    out_i = cfg.witness_batch_position_major(c_pm, d_lens, chars_pm_stride=stride)
    out_p = cfg.witness_batch_planes(c_pm, d_lens, chars_pm_stride=stride, out=cfg.alloc_output_planes(B, dev, stripes=stripes))
    torch.cuda.synchronize()
    assert torch.equal(out_i[2], out_p[2])
    for canonical in (False, True):
        a = cfg.fr_columns(c_pm, d_lens, out_i, b_begin=37, b_count=500, position_major=True, chars_pm_stride=stride, canonical=canonical)
        b = cfg.fr_columns(c_pm, d_lens, out_p, b_begin=37, b_count=500, position_major=True, chars_pm_stride=stride, canonical=canonical)
        torch.cuda.synchronize()
        ok = ((out_i[2][37:537] & 0xff) == 0)
        assert ok.sum() > 400 and torch.equal(a[:, ok], b[:, ok])
