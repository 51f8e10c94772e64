"""hrx_witness_batch_host's three routes (HRX_OPT_HOST_ROUTE): everything through the device, everything on the host cores, and the default — the batch split by string index
between both, in the ratio of the rates the context measured.  The rows must be identical whichever way they were made (src/lib.rs:311-318: host Vecs in, host Vecs out)."""
import os

import numpy as np
import pytest

from oracle_lib import OracleDefs
from test_parity_gpu import CFG_1, CFG_A, _cfg, hra  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("names", [CFG_1, CFG_A], ids=["D1", "D2"])
def test_the_three_host_routes_give_identical_rows(hra, oracle, names):
    from halo2_regex_amd import synth
    M, B = 1024, 16384                      # 2^24 rows: above the split threshold (2^22)
    base_c, base_l = synth.regex1_planted(2048, M - 1, seed=3, stride=M)
    chars, lens = np.tile(base_c, (B // 2048, 1)), np.tile(base_l, B // 2048)
    lens[5], lens[B - 7] = 0, M + 9         # an empty string, a string longer than max_chars_size (status 3) in each part
    chars[11, 40] = 200                     # an undefined transition
    cfg = _cfg(hra, names, M)
    o = OracleDefs.from_files(oracle, names)
    orec, omsk, ost = o.witness_batch(base_c, base_l, M, threads=os.cpu_count() or 1)
    outs = {}
    for name, route in (("device", hra.HOST_ROUTE_DEVICE), ("host", hra.HOST_ROUTE_HOST), ("auto", hra.HOST_ROUTE_AUTO)):
        cfg.set_option(hra.OPT_HOST_ROUTE, route)
        assert cfg.get_option(hra.OPT_HOST_ROUTE) == route
        seen = set()
        for _ in range(7 if name == "auto" else 1):      # AUTO: calls 0-4 measure (device, device, host cores, split, split), then the fastest way
            outs[name] = cfg.witness_batch_host(chars, lens)
            seen.add(cfg.host_route_report()["route"])
            if name == "auto":
                outs.setdefault("auto_all", []).append(outs[name])
        rep = cfg.host_route_report()
        if name == "device":
            assert rep["route"] == 1 and rep["device_strings"] == B and rep["host_strings"] == 0
        elif name == "host":
            assert rep["route"] == 2 and rep["host_strings"] == B and rep["host_threads"] >= 1
        else:       # all three ways ran and were timed; the figures are what the next call picks its way by
            assert seen == {0, 1, 2}
            assert rep["device_strings"] + rep["host_strings"] == B and rep["device_strings"] % 64 == 0
            assert rep["device_alone_ns_per_row"] > 0 and rep["host_alone_ns_per_row"] > 0 and rep["split_ns_per_row"] > 0 and rep["call_ms"] > 0
            best = min(rep["device_alone_ns_per_row"], rep["host_alone_ns_per_row"], rep["split_ns_per_row"])
            assert {0: rep["split_ns_per_row"], 1: rep["device_alone_ns_per_row"], 2: rep["host_alone_ns_per_row"]}[rep["route"]] <= 1.6 * best
    ref = outs["device"]
    ok = (ref[2] & np.uint64(0xff)) == 0
    assert (~ok).sum() >= 2
    for name, got in [("host", outs["host"])] + [("auto call %d" % i, g) for i, g in enumerate(outs["auto_all"])]:
        assert np.array_equal(got[2], ref[2]), name
        assert np.array_equal(got[0][ok], ref[0][ok]) and np.array_equal(got[1][ok], ref[1][ok]), name
    for k in range(2048, B - 2048, 2048):   # ... and they are the oracle's (the copies of the base strings that were not modified above)
        sl = slice(k, k + 2048)
        good = ok[sl]
        assert np.array_equal(ref[0][sl][good], orec[good]) and np.array_equal(ref[1][sl][good], omsk[good])
    cl = cfg.clone()
    assert cl.get_option(hra.OPT_HOST_ROUTE) == hra.HOST_ROUTE_AUTO


def test_host_route_options_are_checked(hra):
    cfg = _cfg(hra, CFG_1, 64)
    with pytest.raises(hra.HrxError):
        cfg.set_option(hra.OPT_HOST_ROUTE, 3)
    with pytest.raises(hra.HrxError):
        cfg.set_option(99, 0)
    cfg.set_option(hra.OPT_HOST_THREADS, 3)
    cfg.set_option(hra.OPT_HOST_PIPELINE, 2)
    assert cfg.get_option(hra.OPT_HOST_THREADS) == 3 and cfg.get_option(hra.OPT_HOST_PIPELINE) == 2 and cfg.get_option(99) == -1
