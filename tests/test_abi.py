"""CPU-side checks of the C-ABI library: it loads, exports exactly what include/hrx.h declares, and its host
logic (parsers of src/defs.rs, table rows of src/table.rs, sharding) agrees with the oracle.  No compute calls."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import halo2_regex_amd as hra
from halo2_regex_amd import AllstrRegexDef, SubstrRegexDef, RegexDefs, RegexVerifyConfig
from oracle_lib import OracleDefs, DFA_DIR, ROOT


def _defs(names):
    return [RegexDefs(AllstrRegexDef.read_from_text(os.path.join(DFA_DIR, a)),
                      [SubstrRegexDef.read_from_text(os.path.join(DFA_DIR, s)) for s in subs]) for a, subs in names]


CFG_A = [["regex1_test_lookup.txt", ["substr1_test_lookup.txt"]], ["regex2_test_lookup.txt", ["substr2_test_lookup.txt"]]]
CFG_3 = [["regex3_test_lookup.txt", ["substr3_test_lookup.txt"]]]


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "hrx.h")).read()
    declared = set(re.findall(r"\b(hrx_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(hra.ABI_SYMBOLS)
    out = subprocess.check_output(["nm", "-D", "--defined-only", hra.LIB_PATH]).decode()
    exported = set(re.findall(r" T (hrx_[a-z0-9_]+)", out))
    assert declared <= exported
    for s in declared:
        assert getattr(hra.lib, s) is not None


def test_rust_binding_declares_every_export():
    """bindings/rust/hrx.rs (uncompiled here: no Rust toolchain in the image) must at least name every entry point of include/hrx.h and nothing else."""
    hdr = open(os.path.join(ROOT, "include", "hrx.h")).read()
    rs = open(os.path.join(ROOT, "bindings", "rust", "hrx.rs")).read()
    declared = set(re.findall(r"\b(hrx_[a-z0-9_]+)\s*\(", hdr))
    bound = set(re.findall(r"pub fn (hrx_[a-z0-9_]+)\s*\(", rs))
    assert declared == bound, (sorted(declared - bound), sorted(bound - declared))


def test_library_is_a_gfx950_hip_build():
    # the product .so carries a gfx950 code object (hipcc --offload-arch=gfx950) and no other target
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o", "--input=" + hra.LIB_PATH],
                         capture_output=True, text=True)
    blob = open(hra.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"witness_kernel" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_90"):
        assert other not in blob


def test_host_constants_and_table_rows_match_oracle(oracle):
    cfg = RegexVerifyConfig.configure(1024, _defs(CFG_A), device=None)
    o = OracleDefs.from_files(oracle, CFG_A)
    assert cfg.num_defs == 2
    assert [cfg.substr_id_offset(d) for d in range(2)] == [1, 2]          # lib.rs:780-783
    assert [cfg.accepted_state(d) for d in range(2)] == [24, 12]
    assert cfg.table_bytes() == (29 + 2 + 13 + 2) * 1024                  # real states + dummy + dead, 1 KiB per row
    for d, (tr, ep) in enumerate(cfg.load()):
        assert np.array_equal(tr, o.table_transition_rows(d))             # table.rs:68-125, file-line order
        assert np.array_equal(ep, o.table_endpoint_rows(d))               # table.rs:126-196
    assert len(cfg.load()[0][0]) == 1 + 2842


def test_parser_edge_cases_follow_defs_rs():
    # duplicate (char,state) key: last line wins but keeps ITS line index (defs.rs:100, table.rs:108)
    txt = "0\n1\n1\n0 0 97\n0 1 98\n0 1 97\n"
    cfg = RegexVerifyConfig.configure(8, [RegexDefs(AllstrRegexDef(txt), [])], device=None)
    rows = cfg.load()[0][0]
    assert [list(map(int, r)) for r in rows] == [[0, 2, 2, 0], [98, 0, 1, 0], [97, 0, 1, 0]]
    # the char column is taken `as u8`: 353 -> 97 (defs.rs:100)
    cfg = RegexVerifyConfig.configure(8, [RegexDefs(AllstrRegexDef("0\n1\n1\n0 1 353\n"), [])], device=None)
    assert [int(x) for x in cfg.load()[0][0][1]] == [97, 0, 1, 0]
    # CRLF line ends and no trailing newline parse like BufRead::lines + split_whitespace
    cfg = RegexVerifyConfig.configure(8, [RegexDefs(AllstrRegexDef("0\r\n1\r\n1\r\n0 1 97"), [])], device=None)
    assert len(cfg.load()[0][0]) == 2
    for bad in ("0\n1\n1\n0 1\n", "0\n1\n1\n0 x 97\n", "0\n\n1\n", "0\n1\n1\n0 -1 97\n"):
        with pytest.raises(hra.HrxError) as e:
            RegexVerifyConfig.configure(8, [RegexDefs(AllstrRegexDef(bad), [])], device=None)
        assert e.value.code == hra.HRX_ERR_PARSE
    with pytest.raises(hra.HrxError) as e:   # substr pair line with one element (defs.rs:238 indexes elements[1])
        RegexVerifyConfig.configure(8, [RegexDefs(AllstrRegexDef("0\n1\n1\n0 1 97\n"), [SubstrRegexDef("4\n0\n7\n0 \n1 \n0\n")])],
                                    device=None)
    assert e.value.code == hra.HRX_ERR_PARSE


def test_bounds_are_rejected_loudly():
    with pytest.raises(hra.HrxError) as e:   # transition to a state above largest_state_val
        RegexVerifyConfig.configure(8, [RegexDefs(AllstrRegexDef("0\n1\n1\n0 5 97\n"), [])], device=None)
    assert e.value.code == hra.HRX_ERR_BOUNDS
    big = "0\n1\n200\n" + "".join("%d %d 97\n" % (s, s) for s in range(201))
    cfg = RegexVerifyConfig.configure(8, [RegexDefs(AllstrRegexDef(big), [])], device=None)   # > LDS: global-table kernels
    assert cfg.table_bytes() == 203 * 1024
    huge = "0\n1\n3000\n0 1 97\n"
    with pytest.raises(hra.HrxError) as e:   # 3003 table rows: beyond the supported table size
        RegexVerifyConfig.configure(8, [RegexDefs(AllstrRegexDef(huge), [])], device=None)
    assert e.value.code == hra.HRX_ERR_BOUNDS
    with pytest.raises(hra.HrxError):
        RegexVerifyConfig.configure(8, [], device=None)


def test_substr_new_matches_text_form(oracle):
    sd = SubstrRegexDef.new(7, 0, 127, {(23, 1), (1, 1)}, [23], [1])
    cfg = RegexVerifyConfig.configure(128, [RegexDefs(AllstrRegexDef.read_from_text(os.path.join(DFA_DIR, "ex_allstr.txt")), [sd])],
                                      device=None)
    o = OracleDefs.from_files(oracle, [["ex_allstr.txt", ["ex_substr_id1.txt"]]])
    assert np.array_equal(cfg.load()[0][0], o.table_transition_rows(0))
    assert np.array_equal(cfg.load()[0][1], o.table_endpoint_rows(0))


def test_shard_range_partitions_the_batch():
    for B in (0, 1, 63, 64, 65, 1000, 65536, 262144):
        for world in (1, 2, 3, 4, 8):
            parts = [hra.shard_range(B, world, r) for r in range(world)]
            assert sum(c for _, c in parts) == B
            pos = 0
            for b, c in parts:
                assert b == pos or c == 0
                pos += c
            assert max(c for _, c in parts) == -(-B // world) if B else True


def test_compute_fails_loudly_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hra.HrxError) as e:
        RegexVerifyConfig.configure(64, _defs(CFG_3), device=0)
    assert e.value.code == hra.HRX_ERR_HIP
    cfg = RegexVerifyConfig.configure(64, _defs(CFG_3), device=None)
    with pytest.raises(hra.HrxError):
        cfg.derive_states(b"from:a@b.com\r\n")
    with pytest.raises(hra.HrxError):
        cfg.witness_batch_host(np.zeros((1, 16), np.uint8), np.zeros(1, np.uint32))


def test_placement_switches_check_their_arguments():
    """hrx_ctx_set_placement (per-context opt-out and budget of the placement walk): argument checks, on a host-only context — no device needed"""
    import ctypes as C
    cfg = RegexVerifyConfig.configure(64, _defs(CFG_3), device=hra.HRX_DEVICE_NONE)
    cfg.set_placement(walk=False)
    cfg.set_placement(walk=True, max_bytes=1 << 30, max_ms=50.0)
    assert hra.lib.hrx_ctx_set_placement(cfg._ctx, 2, 0, C.c_double(0.0)) == hra.HRX_ERR_ARG
    assert hra.lib.hrx_ctx_set_placement(cfg._ctx, hra.PLACE_WALK, 0, C.c_double(-0.5)) == hra.HRX_ERR_ARG
    assert hra.lib.hrx_ctx_set_placement(None, hra.PLACE_OFF, 0, C.c_double(0.0)) == hra.HRX_ERR_ARG
    rep = cfg.last_placement_report()
    assert rep["searched"] == 0 and rep["capped"] == 0


def test_rows_of_one_string_out_of_position_major_host_buffers():
    """hrx_rows_of_string_position_major against the layout's definition (include/hrx.h): blocked by 65536 strings, partial last quad / octet, D up to 5"""
    rng = np.random.default_rng(0)
    for (B, M, D) in [(5, 7, 1), (70000, 9, 2), (130, 64, 3), (65537, 8, 5)]:
        q4, q8 = (M + 3) // 4, (M + 7) // 8
        rec_sm = rng.integers(0, 2 ** 32, size=(B, M, D), dtype=np.uint32)
        msk_sm = rng.integers(0, 2 ** 16, size=(B, M), dtype=np.uint16)
        rp, mp = np.zeros(q4 * B * 4 * D, np.uint32), np.zeros(q8 * B * 8, np.uint16)
        for b in range(0, B, max(1, B // 997)):            # scatter a sample of strings by the formula of include/hrx.h
            k, bl = b // hra.PM_BLOCK, b % hra.PM_BLOCK
            nb = min(hra.PM_BLOCK, B - k * hra.PM_BLOCK)
            for r in range(M):
                for d in range(D):
                    rp[k * hra.PM_BLOCK * q4 * D * 4 + ((r // 4 * D + d) * nb + bl) * 4 + r % 4] = rec_sm[b, r, d]
                mp[k * hra.PM_BLOCK * q8 * 8 + (r // 8 * nb + bl) * 8 + r % 8] = msk_sm[b, r]
            r1, m1 = hra.rows_of_string_position_major(rp, mp, B, M, D, b)
            assert np.array_equal(r1, rec_sm[b]) and np.array_equal(m1, msk_sm[b]), (B, M, D, b)
        with pytest.raises(hra.HrxError):
            hra.rows_of_string_position_major(rp, mp, B, M, D, B)


def test_context_clone_is_an_independent_context_of_the_same_config(oracle):
    """hrx_ctx_clone (RegexVerifyConfig derives Clone, lib.rs:96): the clone outlives its source and the defs handle, carries the per-context switches, computes the same rows"""
    cfg = RegexVerifyConfig.configure(64, _defs(CFG_3), device=hra.HRX_DEVICE_NONE)
    cfg.set_host_threshold(12345)
    c2 = cfg.clone()
    assert c2._ctx.value != cfg._ctx.value and c2.host_threshold() == 12345
    s = b"from:alice@gmail.com\r\n"
    want = cfg.match_substrs(s)
    del cfg
    import gc
    gc.collect()
    got = c2.match_substrs(s)
    assert want.status == got.status and want.status & 0xff == 0 and (want.status >> 8) & 1 == 1     # ok, accept bit of def 0
    for k in ("all_enable_flags", "all_characters", "all_substr_ids", "masked_characters", "states", "substr_ids", "start_enables", "end_enables"):
        assert np.array_equal(getattr(want, k), getattr(got, k)), k
    assert bytes(got.masked_characters[5:20].astype(np.uint8)) == b"alice@gmail.com"        # lib.rs:1318-1331
    import ctypes as C
    out = C.c_void_p()
    assert hra.lib.hrx_ctx_clone(None, hra.HRX_DEVICE_SAME, C.byref(out)) == hra.HRX_ERR_ARG
    assert hra.lib.hrx_ctx_clone(c2._ctx, hra.HRX_DEVICE_SAME, None) == hra.HRX_ERR_ARG


def test_planner_thresholds_of_round_3():
    """Where the chunked launch stops (below 2 / up to 1.75 / up to 1.5 groups of 64 strings per CU at D = 1 / 2 / 3, either input layout,
    position-major outputs, 4096 rows or more) and the one-round rule for small batches (profiles/r03_probes/spec_threshold.txt)."""
    M = 32768
    one = RegexVerifyConfig.configure(M, _defs(CFG_A[:1]), device=None)
    two = RegexVerifyConfig.configure(M, _defs(CFG_A), device=None)
    for lay in (3, 1):
        assert "chunked=32x16 tiles" in one.describe_launch(8192, layout=lay)
        assert "chunked=" in one.describe_launch(32704, layout=lay) and "chunked=" not in one.describe_launch(32768, layout=lay)     # 511 / 512 groups
        assert "chunked=" in two.describe_launch(28672, layout=lay) and "chunked=" not in two.describe_launch(28736, layout=lay)     # 448 / 449 groups
    assert "chunked=" not in one.describe_launch(8192, layout=0) and "chunked=" not in one.describe_launch(8192, layout=2)           # string-major outputs: never
    short = RegexVerifyConfig.configure(2048, _defs(CFG_A[:1]), device=None)
    assert "chunked=" not in short.describe_launch(8192, layout=3)                                                                 # fewer than 4096 rows: never
    # 257-511 groups on the pair-step kernel (76-KiB table: one workgroup per CU): two pairs per workgroup, one round — not one pair and two rounds
    assert short.describe_launch(20480, layout=3).startswith("hrx::witness_pp_kernel grid=160 waves=6 ")
    # 513-768 groups, D = 2 on the WIDE table: three pairs per workgroup cover them in one round
    short2 = RegexVerifyConfig.configure(2048, _defs(CFG_A), device=None)
    assert short2.describe_launch(40000, layout=3).startswith("hrx::witness_pm_kernel<2, false, true, false, false, false> grid=209 waves=9 ")


def test_planner_picks_the_documented_kernel_per_config(monkeypatch):
    """hrx_describe_launch (host-only): which kernel and table format serve which shape on a 256-CU MI355X."""
    from halo2_regex_amd import synth
    cfg = RegexVerifyConfig.configure(1024, _defs(CFG_A[:1]), device=None)
    assert cfg.describe_launch(65536, layout=3).startswith("hrx::witness_pm_kernel<1, false, false, false, false, false> grid=256 waves=12 ring=4 ")   # a full chip: one byte per lookup; walker + loader + finisher wave per pair
    assert cfg.describe_launch(32768, layout=3).startswith("hrx::witness_pp_kernel grid=256 waves=6 ")     # walker slots left empty: one def, 18 byte classes -> the pair-step table (76 KiB), two bytes per lookup
    assert cfg.describe_launch(65536, layout=0).startswith("hrx::witness_split_kernel<1, 32, false> ")
    cfg = RegexVerifyConfig.configure(2048, _defs(CFG_A), device=None)
    assert cfg.describe_launch(32768, layout=1).startswith("hrx::witness_pmd_kernel<2, false, false, false> grid=256 waves=6 ")       # <= 2 groups per CU: one walker per def
    assert cfg.describe_launch(65536, layout=1).startswith("hrx::witness_pm_kernel<2, false, true, false, false, false> ")     # D >= 2: the WIDE table
    d = cfg.describe_launch(1 << 20, layout=1)
    assert d.startswith("hrx::witness_pm_kernel<2, false, true, false, false, false> grid=256 waves=12 ") and d.endswith(" groups=dynamic")   # >= 3 groups per walker pair: drawn from a counter
    assert "groups=dynamic" not in cfg.describe_launch(65536, layout=1)
    cfg3 = RegexVerifyConfig.configure(1024, _defs(CFG_A + CFG_3), device=None)
    assert cfg3.describe_launch(65536, layout=0).startswith("hrx::witness_pm_kernel<3, false, true, false, true, false> ")   # string-major D = 3: lane-direct stores
    assert RegexVerifyConfig.configure(1001, _defs(CFG_A + CFG_3), device=None).describe_launch(65536, layout=0).startswith("hrx::witness_kernel<3, false, false> ")
    # cfg 5: 256 states x 256 symbols = 258 KiB of 4-byte entries -> the 128-KiB HALF table, LDS-resident
    a_txt, sub_txt = synth.random_dfa(256, seed=2, alphabet=np.arange(256, dtype=np.uint8), n_substr_pairs=200)
    cfg = RegexVerifyConfig.configure(4096, [RegexDefs(AllstrRegexDef(a_txt), [SubstrRegexDef(sub_txt)])], device=None)
    d = cfg.describe_launch(65536, layout=3)
    # ... one def: the BYTE table (64 KiB of next-state bytes + the pair tags off the chain) leaves room for a finisher wave and a ring of 3 slots
    assert d.startswith("hrx::witness_pm_kernel<1, false, false, false, false, true> grid=256 waves=12 ring=2 ")
    monkeypatch.setenv("HRX_DEBUG_FLAGS", str(0x8000))          # kDbgNoByte: the 128-KiB HALF table, walker + loader only
    d = cfg.describe_launch(65536, layout=3)
    assert d.startswith("hrx::witness_pm_kernel<1, false, false, true, false, false> grid=256 waves=8 ") and "lds=%d" % (128 * 1024 + 4 * (4096 + 128)) in d
    monkeypatch.delenv("HRX_DEBUG_FLAGS")
    # string-major outputs: the walker/storer kernel on the BYTE table (rows in multiples of 8; else the one-wave global-table walk) ...
    d = cfg.describe_launch(65536, layout=0)
    assert d.startswith("hrx::witness_split_kernel<1, 32, true> grid=256 waves=8 ring=2 ")
    # ... without a BYTE image: the HALF-table position-major kernel into context scratch + the transpose kernel
    monkeypatch.setenv("HRX_DEBUG_FLAGS", str(0x8000))
    d = cfg.describe_launch(65536, layout=0)
    assert d.startswith("hrx::witness_pm_kernel<1, false, false, true, false, false> ") and d.endswith("+ hrx::transpose_pm_to_sm_kernel")
    monkeypatch.delenv("HRX_DEBUG_FLAGS")
    assert RegexVerifyConfig.configure(4097, [RegexDefs(AllstrRegexDef(a_txt), [SubstrRegexDef(sub_txt)])], device=None).describe_launch(65536, layout=0).startswith("hrx::witness_kernel<1, false, true> ")
    # beyond 256 states there is no HALF image: global-table walk
    a_txt, sub_txt = synth.random_dfa(300, seed=2)
    cfg = RegexVerifyConfig.configure(1024, [RegexDefs(AllstrRegexDef(a_txt), [SubstrRegexDef(sub_txt)])], device=None)
    assert cfg.describe_launch(65536, layout=1).startswith("hrx::witness_pm_kernel<1, true, false, false, false, false> ")


def test_readme_quick_start_runs():
    """The README's quick start is executed as written (from the repo root, host-only context)."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "README.md")).read()
    block = re.search(r"```python\n(.*?)```", text, re.S).group(1)
    r = subprocess.run([sys.executable, "-c", block], cwd=root, capture_output=True, text=True, timeout=120, env=dict(os.environ, PYTHONPATH=root))
    assert r.returncode == 0, r.stderr
    assert "b'vitalik'" in r.stdout


def test_header_is_plain_c99():
    """include/hrx.h is the boundary a cgo / bindgen / ctypes consumer reads: it must compile as C (no C++ in it), warnings included."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = '#include "include/hrx.h"\nint main(void) { return hrx_last_error() == 0; }\n'
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", root, "-x", "c", "-"], input=src, capture_output=True, text=True, cwd=root)
    assert r.returncode == 0, r.stderr
