// hrx_kernel_pp.hip — PAIR-STEP position-major kernel (one def): TWO input bytes per dependent LDS lookup.
//
// The position-major walk of hrx_kernel_pm.hip is one dependent `ds_read -> v_and_or -> ds_read` per row and string
// (~50 ns per row whatever the batch size: 54 us of the 86-us headline launch, 1.7 ms for 8192 x 32768-byte strings).
// Small DFAs have few distinct table columns (regex1: 29 states, 18 byte classes), so the walk over TWO bytes fits LDS as a
// table indexed (state, class(c0), class(c1)): 30 x 18^2 x 8 B = 76 KiB (hrx_lane.h, PAIR table).  The dependent chain per
// two rows is one v_mad_u32_u16 + one ds_read_b64, and the entry already holds both rows' finished records:
//   lo = next block address / 8 | substr_id(row 0) << 16 | substr_id(row 1) << 24
//   hi = state(row 0) | state(row 1) << 8 | flags(row 0) << 16 | flags(row 1) << 24          flags = is_start | is_end << 1
//   record(row i) = one v_perm_b32 of (lo, hi).
// The byte -> class translation and the pair index (class0 * C + class1) * 8 are OFF the chain: the loader wave computes
// them while it hands a tile over (it idles on vmcnt otherwise) and the ring slot carries 4 KiB of u16 pair indices next to
// the 4 KiB of raw bytes (which the walker touches only where a reveal mask bit is set or a slow path needs them).
// Same lane algorithm otherwise: per-tile position bitvectors + carry-chain mask scans (hrx_lane.h), same slow paths,
// same buffers and status words as witness_pm_kernel — src/lib.rs:804-888, 339-348, 387-519, 593-764 row for row.
#include <hip/hip_runtime.h>

#include "hrx_device.h"
#include "hrx_walk_pm.h"

namespace hrx {

// next lookup address = (chain word's low 16 bits) * 8 + pair index: the whole VALU part of the dependent chain
__device__ __forceinline__ uint32_t pp_addr(uint32_t lo, uint32_t idx) {
    uint32_t r;
    asm("v_mad_u32_u16 %0, %1, 8, %2" : "=v"(r) : "v"(lo), "v"(idx));
    return r;
}
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c) {   // (the plain expression becomes a v_mad_u64_u32)
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t rotr32(uint32_t x, int s) { return (s & 31) ? __builtin_amdgcn_alignbit(x, x, (uint32_t)(s & 31)) : x; }

struct PpLane {
    uint32_t lo;   // chain word: low 16 bits = LDS byte address / 8 of the block of the current state
    uint32_t cmp;  // a word whose byte 3 is the substr id of the previous row (id-changed bit of the next row)
    uint32_t mx;   // last chain word a LIVE pair produced (reaching the dead block = an undefined transition)
};

// One 64-row tile = 32 pair steps.  iw: the tile's pair indices (u16 x 32).  !FULL: rem = n - t0, mrem = M - 1 - t0 as in
// walk_tile_pm; a pair is masked down to its live rows before the common record / flag code sees it.
template <bool FULL, class Sink>
__device__ __forceinline__ TileBits walk_tile_pp(PpLane &L, const uint32_t (&iw)[16], const WitnessArgs &a, Sink &sink, const int rem, const int mrem,
                                                 uint32_t (&sidq)[16], uint32_t &acc_state, uint32_t &odd_dead) {
    uint32_t st[2] = {0, 0}, enS[3] = {0, 0, 0}, ch[2] = {0, 0};   // enS: end flags one bit up (row p at bit p + 1)
    uint32_t rbuf[4];
    uint32_t lo = L.lo, cmp = L.cmp;
    uint32_t elo = 0, ehi = 0, prev_lo = 0;
    const uint32_t dummy = a.dc[0].dummy_state;   // largest + 1: the padding state AND the state id of the dead block

    auto post = [&](const int k) {   // rows p = 2k and p + 1 from the entry (elo, ehi)
        const int p = 2 * k;
        uint32_t xlo = elo, xhi = ehi;
        if (!FULL) {
            const bool l0 = p < rem, l1 = p + 1 < rem;
            const uint32_t cur = xhi & 0xffu, mid = (xhi >> 8) & 0xffu;
            // rows >= n: lib.rs:404-418 (row n shows s[n], later rows the dummy state; ids and flags 0);
            // end_enable of row M-1 is never assigned (lib.rs:501)
            const uint32_t keep_lo = 0xffffu | (l0 ? 0x00ff0000u : 0u) | (l1 ? 0xff000000u : 0u);
            const uint32_t keep_hi = (l0 ? (p >= mrem ? 0x00010000u : 0x00030000u) : 0u) | (l1 ? (p + 1 >= mrem ? 0x01000000u : 0x03000000u) : 0u);
            const uint32_t s0 = p <= rem ? cur : dummy, s1 = p + 1 <= rem ? mid : dummy;
            if (p == rem) acc_state = cur;                                   // the state at row n (lib.rs:437-457)
            if (p + 1 == rem) { acc_state = mid; odd_dead |= (mid == dummy) ? 1u : 0u; }
            xlo &= keep_lo;
            xhi = (xhi & keep_hi) | s0 | (s1 << 8);
        }
        // compact records: state | substr_id << 16 | flags << 24 — one byte permute each
        rbuf[p & 3] = __builtin_amdgcn_perm(xhi, xlo, 0x06020c04u);
        rbuf[(p & 3) + 1] = __builtin_amdgcn_perm(xhi, xlo, 0x07030c05u);
        if (k & 1) sink.quad(0, p + 1, FULL, mrem, make_uint4(rbuf[0], rbuf[1], rbuf[2], rbuf[3]));
        // start / end flags -> tile bitvectors: hi bit 16 / 17 = row p, bit 24 / 25 = row p + 1
        const uint32_t r0 = rotr32(xhi, 16 - p), r1 = rotr32(xhi, 24 - (p + 1));
        st[p >> 5] |= r0 & (1u << (p & 31));
        enS[(p + 1) >> 5] |= r0 & (1u << ((p + 1) & 31));
        st[(p + 1) >> 5] |= r1 & (1u << ((p + 1) & 31));
        enS[(p + 2) >> 5] |= r1 & (1u << ((p + 2) & 31));
        // id-changed bits, newest row in bit 0 (undone once per word below): substr ids are bytes 2 and 3 of lo
        asm volatile("v_cmp_ne_u32_sdwa vcc, %1, %2 src0_sel:BYTE_2 src1_sel:BYTE_3\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                     "v_cmp_ne_u32_sdwa vcc, %1, %1 src0_sel:BYTE_3 src1_sel:BYTE_2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                     : "+v"(ch[p >> 5]) : "v"(xlo), "v"(cmp) : "vcc");
        cmp = xlo;
        if (k & 1) sidq[k >> 1] = __builtin_amdgcn_perm(xlo, prev_lo, 0x07060302u);   // the tile's substr ids, one byte per row
        prev_lo = xlo;
        sink.row(p);
        sink.row(p + 1);
    };
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const uint32_t idx = (k & 1) ? (iw[k >> 1] >> 16) : (iw[k >> 1] & 0xffffu);
        const uint2 raw = lds_u64(pp_addr(lo, idx));   // delta over two bytes: lib.rs:810, twice
        if (k > 0 && !(a.debug & kDbgPpNoPost)) {
            post(k - 1);
            asm volatile("" : "+v"(st[(2 * k - 1) >> 5]), "+v"(enS[(2 * k) >> 5]), "+v"(ch[(2 * k - 2) >> 5]), "+v"(cmp), "+v"(prev_lo));
            if (!FULL) asm volatile("" : "+v"(acc_state), "+v"(odd_dead));
        }
        __builtin_amdgcn_sched_barrier(0);
        elo = raw.x;
        ehi = raw.y;
        if (FULL) {
            lo = raw.x;
        } else {
            const bool l1 = 2 * k + 1 < rem;
            lo = l1 ? raw.x : lo;           // the chain stops at row n
            L.mx = l1 ? raw.x : L.mx;
            asm volatile("" : "+v"(lo), "+v"(L.mx));
        }
    }
    if (!(a.debug & kDbgPpNoPost)) post(31);
    L.lo = lo;
    L.cmp = cmp;
    if (FULL) L.mx = lo;
    TileBits tb;
    tb.st = (uint64_t)st[0] | ((uint64_t)st[1] << 32);
    tb.en1 = (uint64_t)__builtin_amdgcn_alignbit(enS[1], enS[0], 1) | ((uint64_t)__builtin_amdgcn_alignbit(enS[2], enS[1], 1) << 32);
    tb.ch = (uint64_t)__builtin_bitreverse32(ch[0]) | ((uint64_t)__builtin_bitreverse32(ch[1]) << 32);
    return tb;
}

// Three waves per group of 64 strings, like witness_pm_kernel: walker (chain, records, error paths, status), loader (input
// + byte -> class translation) and FINISHER (reveal masks, fix-ups, masked rows from the raw bytes in the ring slot and the
// 96-byte-per-lane tile summary the walker hands over).  This kernel serves batches that leave walker slots — whole SIMDs —
// empty, so the third wave costs nothing and takes ~a quarter of the walker's tile time off the string's serial chain.
__global__ __launch_bounds__(768) void witness_pp_kernel(const WitnessArgs a, const uint32_t nring) {
    constexpr uint32_t kSlot = (uint32_t)kPpSlotBytes, kRaw = 4096u;   // slot = [pair indices 4 KiB][raw bytes 4 KiB]
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t pairs = (blockDim.x >> 6) / 3u;  // walker waves 0..pairs-1, loader waves pairs..2*pairs-1, finisher waves 2*pairs..3*pairs-1
    const bool is_walker = wave < pairs, is_finisher = wave >= 2u * pairs;
    const uint32_t pair = wave % pairs;
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();

    const uint32_t tab_bytes = a.pair_bytes, lut = a.pair_lut_off, C = a.pair_classes;
    const uint32_t ring_base = tab_bytes + pair * (uint32_t)pp_pair_bytes(nring);
    const uint32_t sum_off = ring_base + nring * kSlot;                                   // the tile summary (kPmSummaryBytes)
    const uint32_t ready_off = sum_off + (uint32_t)kPmSummaryBytes, freed_off = ready_off + 4u;
    const uint32_t freed2_off = ready_off + 8u, sum_ready_off = ready_off + 12u, sum_freed_off = ready_off + 16u;
    const uint32_t M = a.M, B = a.B;
    const uint32_t ntiles = (M + 63u) >> 6;
    uint32_t seq = 0, ready_seen = 0;
    const uint32_t g_first = xcd_slot(blockIdx.x, gridDim.x, (a.debug & kDbgXcdRemap) != 0) * pairs + pair, g_stride = gridDim.x * pairs;
    // the loaders request their pair's first input tile, the walkers their first lengths, BEFORE the table is staged
    uint32_t first_len = M;
    if (is_walker && g_first < a.n_groups) first_len = a.lens[min(g_first * 64u + lane, B - 1u)];
    uint4 first_tile[4];
    if (!is_walker && !is_finisher && g_first < a.n_groups) {
        const bool in_pm0 = (a.layout & 2u) != 0;
        const uint32_t bl = min(g_first * 64u + lane, B - 1u);
        const uint32_t blk0 = (g_first * 64u / kPmBlock) * kPmBlock, nb0 = min(kPmBlock, B - blk0);
        const uint8_t *cptr = in_pm0 ? a.chars + (size_t)blk0 * a.stride + (size_t)(bl - blk0) * 16u : a.chars + (size_t)bl * a.stride;
        const uint32_t row_cap0 = (uint32_t)a.stride - 16u;
        const size_t cmul0 = (a.debug & kDbgInputFromL2) ? (size_t)0 : in_pm0 ? (size_t)nb0 : (size_t)1;
#pragma unroll
        for (uint32_t i = 0; i < 4u; ++i) first_tile[i] = *reinterpret_cast<const uint4 *>(cptr + (size_t)min(16u * i, row_cap0) * cmul0);
    }
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.pair_image);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = threadIdx.x; i < tab_bytes / 16u; i += blockDim.x) dst[i] = src[i];
        if (is_walker && lane == 0) {
            lds_store_u32(ready_off, 0); lds_store_u32(freed_off, 0);
            lds_store_u32(freed2_off, 0); lds_store_u32(sum_ready_off, 0); lds_store_u32(sum_freed_off, 0);
        }
    }
    __syncthreads();

    if (is_finisher) {
        // ================================ finisher ================================
        const uint32_t my_groups = g_first < a.n_groups ? (a.n_groups - g_first + g_stride - 1u) / g_stride : 0u;
        const size_t q8 = (M + 7u) / 8u;
        const bool nt_msk = !(a.debug & (kDbgNoNtStores | kDbgNoNtMasked));
        uint32_t f = 0;
        for (uint32_t j = 0; j < my_groups; ++j) {
            const uint32_t b0 = (g_first + j * g_stride) * 64u, b = b0 + lane;
            const bool active = b < B;
            const uint32_t bc = active ? b : B - 1u;     // lanes beyond the batch shadow the last string (same values, same addresses)
            const uint32_t blk0 = (b0 / kPmBlock) * kPmBlock, nb = min(kPmBlock, B - blk0);
            unsigned char *mp = reinterpret_cast<unsigned char *>(a.masked) + ((size_t)blk0 * q8 + (bc - blk0)) * 16u;
            const size_t mstep = (size_t)nb * 16u;
            MaskCarry mc = {0, 0, 0, 0};
            for (uint32_t tf = 0; tf < ntiles; ++tf, ++f) {
                const uint32_t t0 = tf << 6;
                ring_wait(sum_ready_off, f + 1u);
                const uint4 s0 = lds_u128(sum_off + lane * 16u), s1 = lds_u128(sum_off + 1024u + lane * 16u);
                uint32_t sidq[16], cw[16];
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i) {
                    const uint4 v = lds_u128(sum_off + 2048u + i * 1024u + lane * 16u);
                    sidq[4 * i] = v.x; sidq[4 * i + 1] = v.y; sidq[4 * i + 2] = v.z; sidq[4 * i + 3] = v.w;
                    const uint4 c = lds_u128(ring_base + (f % nring) * kSlot + kRaw + i * 1024u + lane * 16u);
                    cw[4 * i] = c.x; cw[4 * i + 1] = c.y; cw[4 * i + 2] = c.z; cw[4 * i + 3] = c.w;
                }
                ring_post_lds(sum_freed_off, f + 1u);   // (the LDS executes it behind the reads above)
                lds_store_u32(freed2_off, f + 1u);
                TileBits tb;
                tb.st = (uint64_t)s0.x | ((uint64_t)s0.y << 32);
                tb.en1 = (uint64_t)s0.z | ((uint64_t)s0.w << 32);
                tb.ch = (uint64_t)s1.x | ((uint64_t)s1.y << 32);
                const uint32_t n_f = s1.z;
                // ---------------- reveal masks: lib.rs:598-764 ----------------
                TileMasks tm = tile_masks<64>(tb, mc, t0, tile_is_exact(t0, n_f, M), rows_below(t0, n_f));
                if (a.debug & kDbgPpNoMask) { tm.mask = 0; tm.fix = 0; }
                if (!active) tm.fix = 0;
                uint64_t fixm = __ballot(tm.fix != 0);
                while (fixm) {   // an earlier optimistic end_mask = 1 turned out wrong: zero those masked rows (rare with real definitions)
                    const int jj = __ffsll((unsigned long long)fixm) - 1;
                    fixm &= fixm - 1;
                    const uint32_t fs = (uint32_t)__builtin_amdgcn_readlane((int)tm.fix_start, jj);
                    const uint32_t bj = b0 + (uint32_t)jj;
                    for (uint32_t r = fs + lane; r < t0; r += 64u)
                        a.masked[((size_t)blk0 * q8 + (size_t)(r >> 3) * nb + (bj - blk0)) * 8u + (r & 7u)] = 0;
                }
                // ---------------- masked rows of this tile: 8 x 16 B per string, [M/8][B][8] (lib.rs:752-761) ----------------
                const uint32_t mlo = (uint32_t)tm.mask, mhi = (uint32_t)(tm.mask >> 32);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t mbyte = ((k < 4 ? mlo : mhi) >> (8 * (k & 3))) & 0xffu;
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (mbyte) v = masked_octet(cw[2 * k], cw[2 * k + 1], sidq[2 * k], sidq[2 * k + 1], mbyte);
                    if (t0 + (uint32_t)k * 8u < M && !(a.debug & kDbgSkipMasked)) store16(mp + ((size_t)tf * 8u + (size_t)k) * mstep, v, nt_msk);
                }
            }
        }
        return;
    }
    if (!is_walker) {
        // ================================ loader ================================
        // Rolling register prefetch exactly as in witness_pm_kernel (RT tiles ahead, counted vmcnt).  On hand-over the 64
        // bytes of the lane's string go to the slot twice: raw, and as 32 pair indices — class LUT lookups (ds_read_u8,
        // value = class * 8) and one v_mad_u32_u24 per pair, all in the loader's otherwise idle issue slots.
        constexpr uint32_t RT = 10u;
        const bool in_pm = (a.layout & 2u) != 0;
        const uint32_t my_groups = g_first < a.n_groups ? (a.n_groups - g_first + g_stride - 1u) / g_stride : 0u;
        const uint32_t total = my_groups * ntiles;
        const uint32_t row_cap = (uint32_t)a.stride - 16u;
        uint4 buf[RT * 4u];
        auto issue = [&](const uint32_t q, const uint32_t k) {
            const uint32_t g = g_first + (q / ntiles) * g_stride, t = q % ntiles;
            const uint32_t bl = min(g * 64u + lane, B - 1u);
            const uint32_t blk0 = (g * 64u / kPmBlock) * kPmBlock, nb = min(kPmBlock, B - blk0);
            const uint8_t *cptr = in_pm ? a.chars + (size_t)blk0 * a.stride + (size_t)(bl - blk0) * 16u : a.chars + (size_t)bl * a.stride;
            const size_t cmul = (a.debug & kDbgInputFromL2) ? (size_t)0 : in_pm ? (size_t)nb : (size_t)1;   // (ablation: every tile re-reads the hot first lines)
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) buf[k * 4u + i] = *reinterpret_cast<const uint4 *>(cptr + (size_t)min(t * 64u + 16u * i, row_cap) * cmul);
        };
        auto emit = [&](const uint32_t slot, uint4 v, const uint32_t i) {   // chunk i (16 bytes) of the tile
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));   // after the counted wait, not before
            *(__attribute__((address_space(3))) v4u32 *)(uintptr_t)(slot + kRaw + i * 1024u + lane * 16u) = v4u32{v.x, v.y, v.z, v.w};
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            uint32_t o[4] = {0, 0, 0, 0};
            if (!(a.debug & kDbgPpNoTranslate)) {
                uint32_t c[16];
#pragma unroll
                for (int j = 0; j < 4; ++j) {   // all 16 lookups in flight before the first result is used
                    c[4 * j] = lds_u8(lut + (w[j] & 0xffu)); c[4 * j + 1] = lds_u8(lut + ((w[j] >> 8) & 0xffu));
                    c[4 * j + 2] = lds_u8(lut + ((w[j] >> 16) & 0xffu)); c[4 * j + 3] = lds_u8(lut + (w[j] >> 24));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)     // (class0 * C + class1) * 8: the entry's byte offset inside a block
                    o[j] = mad_u24(c[4 * j], C, c[4 * j + 1]) | (mad_u24(c[4 * j + 2], C, c[4 * j + 3]) << 16);
            }
            *(__attribute__((address_space(3))) v4u32 *)(uintptr_t)(slot + i * 1024u + lane * 16u) = v4u32{o[0], o[1], o[2], o[3]};
        };
        if (total > 0) {   // the pair's FIRST tile travels alone (requested before the table staging)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) emit(ring_base, first_tile[i], i);
            ring_post_lds(ready_off, 1u);
        }
#pragma unroll
        for (uint32_t k = 1; k < RT; ++k)
            if (k < total) issue(k, k);
        if (RT < total) issue(RT, 0);
        for (uint32_t s0 = 0; s0 < total; s0 += RT) {
#pragma unroll
            for (uint32_t k = 0; k < RT; ++k) {
                const uint32_t sq = s0 + k;
                if (sq < total && sq != 0u) {
                    if (sq >= nring) {
                        ring_wait(freed_off, sq - nring + 1u);     // the walker is done with this slot
                        ring_wait(freed2_off, sq - nring + 1u);    // ... and the finisher with its raw bytes
                    }
                    const uint32_t slot = ring_base + (sq % nring) * kSlot;
                    // tile sq was requested RT tiles ago; RT-1 younger tiles (4 loads each) may still be in flight
                    if (sq + RT <= total) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tail of the sequence: nothing younger is being issued
#pragma unroll
                    for (uint32_t i = 0; i < 4u; ++i) emit(slot, buf[k * 4u + i], i);
                    ring_post_lds(ready_off, sq + 1u);
                    if (sq + RT < total) issue(sq + RT, k);
                }
            }
        }
        return;
    }

    // ================================ walker ================================
    const uint32_t blk8 = a.pair_blk_bytes >> 3;                       // block size in 8-byte units
    const uint32_t dead8 = a.dc[0].dummy_state * blk8;                 // the dead block (state id largest + 1) sits behind the real states
    for (uint32_t g = g_first; g < a.n_groups; g += g_stride) {
        const uint32_t b0 = g * 64u;
        const uint32_t b = b0 + lane;
        const bool active = b < B;
        const uint32_t n_raw = g == g_first ? first_len : a.lens[min(b, B - 1u)];   // lanes beyond the batch: exact shadows of string B - 1 (hrx_kernel_pm.hip)
        const bool badlen = n_raw > M;
        const uint32_t n = badlen ? M : n_raw;
        const uint32_t min_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_min_u32(n));
        PpLane L;
        L.lo = a.dc[0].first_state * blk8;   // states[0] = first_state_val: lib.rs:807 (byte 3 = 0: no substr id before row 0)
        L.cmp = 0;
        L.mx = L.lo;
        uint32_t dead = 0, err_pos = 0, err_state = 0, err_char = 0;
        uint32_t acc_state = a.dc[0].first_state;  // n == 0
        const uint32_t bc = active ? b : B - 1u;  // idle lanes shadow the last string: they store the same values to the same addresses
        const uint32_t blk0 = (b0 / kPmBlock) * kPmBlock, nb = min(kPmBlock, B - blk0);   // the group's block of the position-major buffers
        const size_t q4 = (M + 3u) / 4u;
        unsigned char *rp = reinterpret_cast<unsigned char *>(a.records) + ((size_t)blk0 * q4 + (bc - blk0)) * 16u;
        const size_t rstep = (size_t)nb * 16u;   // one quad of rows further: [M/4][1][nb][4]
        uint4 pend[8];                           // (the sink's slot for masked rows in flight: unused here, they are the finisher's)
        const size_t poff1[1] = {0};             // (one def: its plane is the buffer)
#pragma unroll
        for (int k = 0; k < 8; ++k) pend[k] = make_uint4(0, 0, 0, 0);

        for (uint32_t t = 0; t < ntiles; ++t, ++seq) {
            const uint32_t t0 = t << 6;
            const uint32_t slot = ring_base + (seq % nring) * kSlot;
            ring_wait_seen(ready_off, seq + 1u, ready_seen);
            const uint32_t sf_seen = lds_vol_u32(sum_freed_off);     // (looked at when the tile is walked: it arrives with the tile's pair indices)
            uint32_t iw[16];
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) {
                const uint4 v = lds_u128(slot + i * 1024u + lane * 16u);
                iw[4 * i] = v.x; iw[4 * i + 1] = v.y; iw[4 * i + 2] = v.z; iw[4 * i + 3] = v.w;
            }
            const uint32_t lo_start = L.lo;
            uint32_t sidq[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) sidq[i] = 0;
            uint32_t odd_dead = 0;
            const bool full = (t0 + 64u < min_n);
            GlobalSink<1, false> sink{rp, poff1, rstep, rstep, 0, !(a.debug & kDbgSkipRecords), !(a.debug & (kDbgNoNtStores | kDbgNoNtRecords)), !(a.debug & (kDbgNoNtStores | kDbgNoNtMasked)),
                                      pend, rp, rstep, false, {}};
            TileBits tb;
            if (full) tb = walk_tile_pp<true>(L, iw, a, sink, 0, 0, sidq, acc_state, odd_dead);
            else tb = walk_tile_pp<false>(L, iw, a, sink, (int)n - (int)t0, (int)M - 1 - (int)t0, sidq, acc_state, odd_dead);
            rp = sink.rp;

            // ---------------- undefined transition (lib.rs:817): rare slow path, re-walk the tile one byte at a time ----------------
            const bool newly = !dead && ((L.mx & 0xffffu) == dead8 || odd_dead != 0);
            if (__any(newly)) {
                if (newly) {
                    uint32_t at8 = lo_start & 0xffffu;
                    const uint32_t live_rows = n > t0 ? min(n - t0, 64u) : 0u;
                    for (uint32_t p = 0; p < live_rows; ++p) {
                        const uint32_t c = smem[slot + kRaw + (p >> 4) * 1024u + lane * 16u + (p & 15u)];
                        const uint32_t hi = lds_u32(at8 * 8u + (uint32_t)smem[lut + c] * C + 4u);   // entry (class(c), class 0): its first row is the step over c
                        const uint32_t mid = (hi >> 8) & 0xffu;
                        if (mid == a.dc[0].dummy_state) {
                            err_pos = t0 + p;
                            err_state = hi & 0xffu;
                            err_char = c;
                            break;
                        }
                        at8 = mid * blk8;
                    }
                    dead = 1;
                }
            }
            // ---------------- accept state when n == M: row n does not exist, s[n] is the live state ----------------
            if (!full && n == t0 + 64u && t + 1 == ntiles) acc_state = lds_u32((L.lo & 0xffffu) * 8u + 4u) & 0xffu;
            ring_post_lds(freed_off, seq + 1u);   // done with the slot's pair indices and (slow path) raw bytes; the finisher frees its own view
            // ---------------- hand the tile over to the finisher wave: bitvectors, substr-id bytes, the string's length ----------------
            ring_wait_seen(sum_freed_off, seq, sf_seen);   // it has consumed the previous tile's summary (one summary area per pair)
            typedef __attribute__((address_space(3))) v4u32 lds_v4u32;
            *(lds_v4u32 *)(uintptr_t)(sum_off + lane * 16u) = v4u32{(uint32_t)tb.st, (uint32_t)(tb.st >> 32), (uint32_t)tb.en1, (uint32_t)(tb.en1 >> 32)};
            *(lds_v4u32 *)(uintptr_t)(sum_off + 1024u + lane * 16u) = v4u32{(uint32_t)tb.ch, (uint32_t)(tb.ch >> 32), n, 0u};
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i)
                *(lds_v4u32 *)(uintptr_t)(sum_off + 2048u + i * 1024u + lane * 16u) = v4u32{sidq[4 * i], sidq[4 * i + 1], sidq[4 * i + 2], sidq[4 * i + 3]};
            ring_post_lds(sum_ready_off, seq + 1u);
            ready_seen = lds_vol_u32(ready_off);      // the next tile's look at the loader's counter
        }
        // ---------------- per-string status ----------------
        if (active) {
            uint64_t sw;
            if (badlen) sw = kStatusBadLength;
            else if (dead) sw = status_invalid(0u, err_pos, err_state, err_char);
            else sw = status_ok(acc_state == a.dc[0].accepted_state ? 1u : 0u);
            a.status[b] = sw;
        }
    }
}

hipError_t launch_witness_pp(const WitnessArgs &a, const LaunchInfo &li, hipStream_t stream) {
    if (a.D != 1 || !a.pair_image || !(a.layout & 1u)) return hipErrorInvalidValue;
    auto k = witness_pp_kernel;
    static std::atomic<size_t> granted[64];  // per device: the attribute is set on the current device's function
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipError_t e = ensure_lds(k, granted[dev & 63], li.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(li.grid), dim3(64 * li.waves_per_wg), li.lds_bytes, stream, a, (uint32_t)li.nslots);
    return hipGetLastError();
}

}  // namespace hrx
