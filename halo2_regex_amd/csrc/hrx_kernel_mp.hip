// hrx_kernel_mp.hip — the combine step of multi-pass configs (more than kMaxDefsPerPass RegexDefs, src/lib.rs:112 is a Vec of
// any length).  The reference's per-row quantities that need EVERY def are sums over the defs (lib.rs:467-471 Sum(substr_id),
// :494-498 Sum(is_start), :501-519 Sum(is_end)); everything per def (state, substr_id, start_enable, end_enable) is final when
// its group's walk has written it.  One lane per string, 64 rows per step:
//   * the groups' finished records (position-major, 16 bytes = 4 rows per lane and load, coalesced) are copied into the caller's
//     records buffer at the def's place (position-major [M/4][D][nb][4] or string-major [B][pitch][D]);
//   * substr ids and flag counts are summed byte-wise (four rows per dword), turned into the tile's ST / EN / id-changed
//     bitvectors, and the reveal mask is the same carry-chain scan every other kernel uses (hrx_lane.h tile_masks, optimistic
//     end-mask protocol and fix-ups included);
//   * two defs flagging the same row (out of contract) is found from the counts, lowest row first;
//   * the groups' status words merge in def order: the lowest def's undefined transition wins (lib.rs:806-817), then the
//     overlap, else the accept bits side by side.
// Pure streaming: 4 D + 1 bytes read and 4 D + 2 written per row.
#include <hip/hip_runtime.h>

#include "hrx_device.h"
#include "hrx_walk_pm.h"

namespace hrx {

template <bool SM>
__global__ __launch_bounds__(256) void witness_combine_kernel(const CombineArgs a) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t B = a.B, M = a.M, D = a.D;
    const uint32_t b0 = (blockIdx.x * 4u + wave) * 64u;
    if (b0 >= B) return;
    const uint32_t b = b0 + lane;
    const bool active = b < B;
    const uint32_t bc = active ? b : B - 1u;     // lanes beyond the batch shadow the last string: same values to the same addresses
    const uint32_t n_raw = a.lens[bc];
    const bool badlen = n_raw > M;
    const uint32_t n = badlen ? M : n_raw;
    const uint32_t blk0 = (b0 / kPmBlock) * kPmBlock, nb = min(kPmBlock, B - blk0), bl = bc - blk0;
    const size_t q4 = (M + 3u) / 4u, q8 = (M + 7u) / 8u;
    const bool in_pm = (a.layout & 2u) != 0;
    const uint8_t *cptr = in_pm ? a.chars + (size_t)blk0 * a.stride + (size_t)bl * 16u : a.chars + (size_t)bc * a.stride;
    const size_t cmul = in_pm ? (size_t)nb : (size_t)1;
    const uint32_t row_cap = (uint32_t)a.stride - 16u;
    MaskCarry mc = {0, 0, 0, 0};
    uint32_t sum_prev = 0, ov_row = 0xffffffffu;
    const uint32_t ntiles = (M + 63u) >> 6;
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));

    for (uint32_t t = 0; t < ntiles; ++t) {
        const uint32_t t0 = t << 6;
        uint32_t sidq[16], stq[16], enq[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) sidq[q] = stq[q] = enq[q] = 0;
        uint32_t d = 0;
        for (uint32_t g = 0; g < a.G; ++g) {
            const uint32_t Dg = a.gD[g];
            for (uint32_t ld = 0; ld < Dg; ++ld, ++d) {
                const uint32_t *gp = a.grec[g] + ((size_t)blk0 * q4 * Dg + bl) * 4u;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const size_t row4 = (size_t)(t0 >> 2) + q;
                    if (row4 < q4) {   // quads that start at or beyond row M do not exist
                        const v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4 *>(gp + (row4 * Dg + ld) * nb * 4u));
                        if (SM) {
                            // string-major records [B][pitch][D]: this def's cell of each of the four rows
                            uint32_t *rp = a.records + ((size_t)bc * a.rec_pitch + row4 * 4u) * D + d;
                            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                            for (uint32_t i = 0; i < 4u; ++i)
                                if (row4 * 4u + i < M) rp[(size_t)i * D] = w[i];
                        } else {
                            __builtin_nontemporal_store(v, reinterpret_cast<v4 *>(a.records + ((size_t)blk0 * q4 * D + (row4 * D + d) * nb + bl) * 4u));
                        }
                        // byte 2 of a record = substr_id, byte 3 = start_enable | end_enable << 1: four rows per dword
                        const uint32_t s4 = __builtin_amdgcn_perm(v.y, v.x, 0x0c0c0602u) | __builtin_amdgcn_perm(v.w, v.z, 0x06020c0cu);
                        const uint32_t f4 = __builtin_amdgcn_perm(v.y, v.x, 0x0c0c0703u) | __builtin_amdgcn_perm(v.w, v.z, 0x07030c0cu);
                        sidq[q] += s4;                          // sums <= 255 (finalize_defs)
                        stq[q] += f4 & 0x01010101u;             // <= 32 defs
                        enq[q] += (f4 >> 1) & 0x01010101u;
                    }
                }
            }
        }
        // the tile's bitvectors (hrx_lane.h TileBits)
        uint64_t st = 0, en1 = 0, ch = 0;
        uint32_t cand_s = 0xffffffffu, cand_e = 0xffffffffu;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            st |= (uint64_t)nonzero_bytes4(stq[q]) << (4 * q);
            en1 |= (uint64_t)nonzero_bytes4(enq[q]) << (4 * q);
            const uint32_t os = (stq[q] + 0x7e7e7e7eu) & 0x80808080u, oe = (enq[q] + 0x7e7e7e7eu) & 0x80808080u;   // a count of 2 or more
            if (os && cand_s == 0xffffffffu) cand_s = t0 + 4u * q + ((uint32_t)__builtin_ctz(os) >> 3);
            if (oe && cand_e == 0xffffffffu) cand_e = t0 + 4u * q + ((uint32_t)__builtin_ctz(oe) >> 3) + 1u;
            const uint32_t x = sidq[q], y = (x << 8) | (q ? (sidq[q - 1] >> 24) : sum_prev);
            ch |= (uint64_t)nonzero_bytes4(x ^ y) << (4 * q);
        }
        sum_prev = sidq[15] >> 24;
        if (ov_row == 0xffffffffu) ov_row = min(cand_s, cand_e);
        TileBits tb{st, en1, ch};
        TileMasks tm = tile_masks<64>(tb, mc, t0, tile_is_exact(t0, n, M), rows_below(t0, n));
        if (!active) tm.fix = 0;
        uint64_t fixm = __ballot(tm.fix != 0);
        while (fixm) {   // an earlier optimistic end_mask = 1 turned out wrong: zero those masked rows (rare)
            const int j = __ffsll((unsigned long long)fixm) - 1;
            fixm &= fixm - 1;
            const uint32_t fs = (uint32_t)__builtin_amdgcn_readlane((int)tm.fix_start, j);
            const uint32_t bj = b0 + (uint32_t)j;
            for (uint32_t r = fs + lane; r < t0; r += 64u)
                a.masked[SM ? (size_t)bj * a.msk_pitch + r : ((size_t)blk0 * q8 + (size_t)(r >> 3) * nb + (bj - blk0)) * 8u + (r & 7u)] = 0;
        }
        // masked rows (lib.rs:752-761): the string's raw bytes of this tile, 16 at a time (bytes at or beyond n are masked away)
        uint32_t cw[16];
#pragma unroll
        for (uint32_t i = 0; i < 4u; ++i) {
            const uint4 c = *reinterpret_cast<const uint4 *>(cptr + (size_t)min(t0 + 16u * i, row_cap) * cmul);
            const bool have = t0 + 16u * i <= row_cap;   // chunks beyond the stride hold no row < n
            cw[4 * i] = have ? c.x : 0u; cw[4 * i + 1] = have ? c.y : 0u; cw[4 * i + 2] = have ? c.z : 0u; cw[4 * i + 3] = have ? c.w : 0u;
        }
        const uint32_t mlo = (uint32_t)tm.mask, mhi = (uint32_t)(tm.mask >> 32);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t row8 = t0 + 8u * k;
            if (row8 < M) {
                const uint32_t mbyte = ((k < 4 ? mlo : mhi) >> (8 * (k & 3))) & 0xffu;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (mbyte) v = masked_octet(cw[2 * k], cw[2 * k + 1], sidq[2 * k], sidq[2 * k + 1], mbyte);
                if (SM) {
                    uint16_t *mp = a.masked + (size_t)bc * a.msk_pitch + row8;
                    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (uint32_t i = 0; i < 8u; ++i)
                        if (row8 + i < M) mp[i] = (uint16_t)(w[i >> 1] >> (16u * (i & 1u)));
                } else {
                    __builtin_nontemporal_store(v4{v.x, v.y, v.z, v.w}, reinterpret_cast<v4 *>(a.masked + ((size_t)blk0 * q8 + (size_t)(row8 >> 3) * nb + bl) * 8u));
                }
            }
        }
    }
    if (!active) return;
    // ---------------- merged status word ----------------
    uint64_t sw = 0;
    bool done = false;
    if (badlen) { sw = kStatusBadLength; done = true; }
    uint32_t accept = 0;
    for (uint32_t g = 0; g < a.G && !done; ++g) {
        const uint64_t s = a.gstatus[g][b];
        const uint32_t code = (uint32_t)(s & 0xffu);
        if (code == kStatusInvalidTransition) {   // groups are in def order and each reports its lowest def: the first one wins (lib.rs:806)
            sw = (s & ~0xff00ull) | ((((s >> 8) & 0xffu) + a.gfirst[g]) << 8);
            done = true;
        } else if (code == kStatusOk) {
            accept |= (uint32_t)((s >> 8) & 0xffu) << a.gfirst[g];
        }   // (a group-internal flag overlap is part of the combined overlap row below; its accept bits are not needed then)
    }
    if (!done) sw = ov_row != 0xffffffffu ? status_overlap(ov_row) : status_ok(accept);
    a.status[b] = sw;
}

// Summary mode (position-major outputs): the passes have written their record planes into the caller's buffer themselves and
// left, per tile and string, 80 bytes of summary (ST / EN bitvectors of the group's defs + one substr-id byte per row).  The
// combine reads G x 1.25 + 1 bytes per row and writes the 2 bytes of masked rows: a config of D defs then costs about
// 4 D + G (1 + 1.25) + 3 bytes per row against 4 D + 3 for a single launch.  Two defs of DIFFERENT groups flagging one row
// show as overlapping bits here; two defs of one group were caught by that group's own walk (its status word says so).
// GP > 0: exactly GP groups, and the NEXT tile's loads (5 GP summary pieces + 4 input pieces per lane) are in flight while a tile is
// processed — without that every tile waited a full memory round trip: 290 us for 65536 x 2048 rows at D = 5 (0.74 GB: 2.5 TB/s).
template <int GP>
__global__ __launch_bounds__(256) void witness_combine_summary_kernel(const CombineArgs a) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t B = a.B, M = a.M;
    const uint32_t b0 = (blockIdx.x * 4u + wave) * 64u;
    if (b0 >= B) return;
    const uint32_t b = b0 + lane;
    const bool active = b < B;
    const uint32_t bc = active ? b : B - 1u;
    const uint32_t n_raw = a.lens[bc];
    const bool badlen = n_raw > M;
    const uint32_t n = badlen ? M : n_raw;
    const uint32_t blk0 = (b0 / kPmBlock) * kPmBlock, nb = min(kPmBlock, B - blk0), bl = bc - blk0;
    const size_t q8 = (M + 7u) / 8u;
    const bool in_pm = (a.layout & 2u) != 0;
    const uint8_t *cptr = in_pm ? a.chars + (size_t)blk0 * a.stride + (size_t)bl * 16u : a.chars + (size_t)bc * a.stride;
    const size_t cmul = in_pm ? (size_t)nb : (size_t)1;
    const uint32_t row_cap = (uint32_t)a.stride - 16u;
    MaskCarry mc = {0, 0, 0, 0};
    uint32_t sum_prev = 0, ov_row = 0xffffffffu;
    const uint32_t ntiles = (M + 63u) >> 6;
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    constexpr int GPn = GP > 0 ? GP : 1;
    uint4 pre_s[GPn][5], pre_c[4];
    auto prefetch = [&](const uint32_t t) {
        if constexpr (GP > 0) {
#pragma unroll
            for (int g = 0; g < GP; ++g) {
                const uint4 *sp = reinterpret_cast<const uint4 *>(a.gsummary[g]) + ((size_t)t * 5u * B + bc);
#pragma unroll
                for (uint32_t i = 0; i < 5u; ++i) pre_s[g][i] = sp[(size_t)i * B];
            }
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) pre_c[i] = *reinterpret_cast<const uint4 *>(cptr + (size_t)min((t << 6) + 16u * i, row_cap) * cmul);
        }
    };
    if (ntiles) prefetch(0);
    for (uint32_t t = 0; t < ntiles; ++t) {
        const uint32_t t0 = t << 6;
        uint32_t sidq[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) sidq[q] = 0;
        uint64_t st = 0, en1 = 0, ov_st = 0, ov_en = 0;
        uint4 cur_c[4];
        if constexpr (GP > 0) {
            uint4 cur_s[GPn][5];
#pragma unroll
            for (int g = 0; g < GP; ++g)
#pragma unroll
                for (uint32_t i = 0; i < 5u; ++i) cur_s[g][i] = pre_s[g][i];
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) cur_c[i] = pre_c[i];
            if (t + 1u < ntiles) prefetch(t + 1u);
#pragma unroll
            for (int g = 0; g < GP; ++g) {
                const uint4 h = cur_s[g][0];
                const uint64_t gst = (uint64_t)h.x | ((uint64_t)h.y << 32), gen = (uint64_t)h.z | ((uint64_t)h.w << 32);
                ov_st |= st & gst;
                ov_en |= en1 & gen;
                st |= gst;
                en1 |= gen;
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i) {
                    const uint4 v = cur_s[g][i + 1u];
                    sidq[4 * i] += v.x; sidq[4 * i + 1] += v.y; sidq[4 * i + 2] += v.z; sidq[4 * i + 3] += v.w;
                }
            }
        }
        for (uint32_t g = 0; GP == 0 && g < a.G; ++g) {
            const uint4 *sp = reinterpret_cast<const uint4 *>(a.gsummary[g]) + ((size_t)t * 5u * B + bc);
            const uint4 h = sp[0];
            const uint64_t gst = (uint64_t)h.x | ((uint64_t)h.y << 32), gen = (uint64_t)h.z | ((uint64_t)h.w << 32);
            ov_st |= st & gst;
            ov_en |= en1 & gen;
            st |= gst;
            en1 |= gen;
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) {
                const uint4 v = sp[(size_t)(i + 1u) * B];
                sidq[4 * i] += v.x; sidq[4 * i + 1] += v.y; sidq[4 * i + 2] += v.z; sidq[4 * i + 3] += v.w;   // byte sums <= 255 (finalize_defs)
            }
        }
        if (ov_row == 0xffffffffu) {
            if (ov_st) ov_row = t0 + (uint32_t)ctz64(ov_st);
            if (ov_en) ov_row = min(ov_row, t0 + (uint32_t)ctz64(ov_en) + 1u);
        }
        uint64_t ch = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint32_t x = sidq[q], y = (x << 8) | (q ? (sidq[q - 1] >> 24) : sum_prev);
            ch |= (uint64_t)nonzero_bytes4(x ^ y) << (4 * q);
        }
        sum_prev = sidq[15] >> 24;
        TileBits tb{st, en1, ch};
        TileMasks tm = tile_masks<64>(tb, mc, t0, tile_is_exact(t0, n, M), rows_below(t0, n));
        if (!active) tm.fix = 0;
        uint64_t fixm = __ballot(tm.fix != 0);
        while (fixm) {   // an earlier optimistic end_mask = 1 turned out wrong: zero those masked rows (rare)
            const int j = __ffsll((unsigned long long)fixm) - 1;
            fixm &= fixm - 1;
            const uint32_t fs = (uint32_t)__builtin_amdgcn_readlane((int)tm.fix_start, j);
            const uint32_t bj = b0 + (uint32_t)j;
            for (uint32_t r = fs + lane; r < t0; r += 64u)
                a.masked[((size_t)blk0 * q8 + (size_t)(r >> 3) * nb + (bj - blk0)) * 8u + (r & 7u)] = 0;
        }
        uint32_t cw[16];
#pragma unroll
        for (uint32_t i = 0; i < 4u; ++i) {
            uint4 c;
            if constexpr (GP > 0) c = cur_c[i];
            else c = *reinterpret_cast<const uint4 *>(cptr + (size_t)min(t0 + 16u * i, row_cap) * cmul);
            const bool have = t0 + 16u * i <= row_cap;
            cw[4 * i] = have ? c.x : 0u; cw[4 * i + 1] = have ? c.y : 0u; cw[4 * i + 2] = have ? c.z : 0u; cw[4 * i + 3] = have ? c.w : 0u;
        }
        const uint32_t mlo = (uint32_t)tm.mask, mhi = (uint32_t)(tm.mask >> 32);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t row8 = t0 + 8u * k;
            if (row8 < M) {
                const uint32_t mbyte = ((k < 4 ? mlo : mhi) >> (8 * (k & 3))) & 0xffu;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (mbyte) v = masked_octet(cw[2 * k], cw[2 * k + 1], sidq[2 * k], sidq[2 * k + 1], mbyte);
                __builtin_nontemporal_store(v4{v.x, v.y, v.z, v.w}, reinterpret_cast<v4 *>(a.masked + ((size_t)blk0 * q8 + (size_t)(row8 >> 3) * nb + bl) * 8u));
            }
        }
    }
    if (!active) return;
    uint64_t sw = 0;
    bool done = false;
    if (badlen) { sw = kStatusBadLength; done = true; }
    uint32_t accept = 0;
    for (uint32_t g = 0; g < a.G && !done; ++g) {
        const uint64_t s = a.gstatus[g][b];
        const uint32_t code = (uint32_t)(s & 0xffu);
        if (code == kStatusInvalidTransition) {
            sw = (s & ~0xff00ull) | ((((s >> 8) & 0xffu) + a.gfirst[g]) << 8);
            done = true;
        } else if (code == kStatusFlagOverlap) {
            ov_row = min(ov_row, (uint32_t)(s >> 40));   // two defs of this group flag the same row
        } else if (code == kStatusOk) {
            accept |= (uint32_t)((s >> 8) & 0xffu) << a.gfirst[g];
        }
    }
    if (!done) sw = ov_row != 0xffffffffu ? status_overlap(ov_row) : status_ok(accept);
    a.status[b] = sw;
}

// The status words of a multi-pass config whose LAST pass merged the groups' summaries itself (WitnessArgs::merge_G): what the tail of the
// combine kernel does, one thread per string.
__global__ __launch_bounds__(256) void witness_merge_status_kernel(const CombineArgs a) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    uint64_t sw = 0;
    bool done = false;
    if (a.lens[b] > a.M) { sw = kStatusBadLength; done = true; }
    uint32_t accept = 0, ov_row = a.merge_ov[b];
    for (uint32_t g = 0; g < a.G && !done; ++g) {
        const uint64_t s = a.gstatus[g][b];
        const uint32_t code = (uint32_t)(s & 0xffu);
        if (code == kStatusInvalidTransition) {          // the lowest def's: the reference walks def by def (lib.rs:806)
            sw = (s & ~0xff00ull) | ((((s >> 8) & 0xffu) + a.gfirst[g]) << 8);
            done = true;
        } else if (code == kStatusFlagOverlap) {
            ov_row = min(ov_row, (uint32_t)(s >> 40));   // two defs of this group flag the same row
        } else if (code == kStatusOk) {
            accept |= (uint32_t)((s >> 8) & 0xffu) << a.gfirst[g];
        }
    }
    if (!done) sw = ov_row != 0xffffffffu ? status_overlap(ov_row) : status_ok(accept);
    a.status[b] = sw;
}

hipError_t launch_merge_status(const CombineArgs &a, hipStream_t stream) {
    if (a.B == 0) return hipSuccess;
    hipLaunchKernelGGL(witness_merge_status_kernel, dim3((a.B + 255u) / 256u), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_combine(const CombineArgs &a, hipStream_t stream) {
    const uint32_t groups64 = (a.B + 63u) / 64u;
    const dim3 grid((groups64 + 3u) / 4u), block(256);
    if (a.gsummary[0]) {
        if (a.G == 2) hipLaunchKernelGGL(witness_combine_summary_kernel<2>, grid, block, 0, stream, a);
        else if (a.G == 3) hipLaunchKernelGGL(witness_combine_summary_kernel<3>, grid, block, 0, stream, a);
        else if (a.G == 4) hipLaunchKernelGGL(witness_combine_summary_kernel<4>, grid, block, 0, stream, a);
        else hipLaunchKernelGGL(witness_combine_summary_kernel<0>, grid, block, 0, stream, a);
    }
    else if (a.layout & 1u) hipLaunchKernelGGL(witness_combine_kernel<false>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(witness_combine_kernel<true>, grid, block, 0, stream, a);
    return hipGetLastError();
}

}  // namespace hrx
