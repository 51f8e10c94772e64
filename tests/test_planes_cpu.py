"""The record-planes / row-stripes layouts of include/hrx.h (hrx_witness_batch_device_planes) on the host: the Python inverse (planes_to_string_major) and the C gather
(hrx_rows_of_string_planes) against the layout's definition, built here index by index — blocks of 65536 strings, odd quad counts, two stripes."""
import numpy as np
import pytest

import halo2_regex_amd as hra


@pytest.mark.parametrize("B,M,D,R", [(300, 203, 3, 1), (70000, 64, 2, 1), (300, 203, 1, 2), (70000, 36, 1, 2), (5, 4, 1, 2), (5, 5, 1, 2)])
def test_planes_and_row_stripes_layout(B, M, D, R):
    rng = np.random.default_rng(B + M)
    rec = rng.integers(0, 2 ** 32, (B, M, D), dtype=np.uint64).astype(np.uint32)
    msk = rng.integers(0, 65536, (B, M)).astype(np.uint16)
    q4, q8 = (M + 3) // 4, (M + 7) // 8
    slots = (q4 + R - 1) // R
    sz_p, sz_m = hra.C.c_size_t(0), hra.C.c_size_t(0)
    hra.lib.hrx_position_major_stripe_sizes(B, M, R, hra.C.byref(sz_p), hra.C.byref(sz_m))
    assert sz_p.value == slots * B * 4 and sz_m.value == q8 * B * 8
    planes = [np.zeros(slots * B * 4, np.uint32) for _ in range(D * R)]
    mpm = np.zeros(q8 * B * 8, np.uint16)
    for k0 in range(0, B, hra.PM_BLOCK):
        nb = min(hra.PM_BLOCK, B - k0)
        for d in range(D):
            for q in range(q4):      # include/hrx.h: quad q of def d lies in buffer (q % R) * D + d at slot q / R of its block
                rows = min(4, M - 4 * q)
                v = planes[(q % R) * D + d][k0 * slots * 4:(k0 + nb) * slots * 4].reshape(slots, nb, 4)
                v[q // R, :, :rows] = rec[k0:k0 + nb, 4 * q:4 * q + rows, d]
        mv = mpm[k0 * q8 * 8:(k0 + nb) * q8 * 8].reshape(q8, nb, 8)
        for o in range(q8):
            rows = min(8, M - 8 * o)
            mv[o, :, :rows] = msk[k0:k0 + nb, 8 * o:8 * o + rows]
    r2, m2 = hra.planes_to_string_major(planes, mpm, B, M, D=D)
    assert np.array_equal(r2, rec) and np.array_equal(m2, msk)
    for b in (0, B - 1, B // 2, min(B - 1, 65536)):
        r1, m1 = hra.rows_of_string_planes(planes, mpm, B, M, b, D=D)
        assert np.array_equal(r1, rec[b]) and np.array_equal(m1, msk[b])
    with pytest.raises(hra.HrxError):
        hra.rows_of_string_planes(planes + planes, mpm, B, M, 0, D=D)
