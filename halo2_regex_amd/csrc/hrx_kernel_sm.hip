// hrx_kernel_sm.hip — gfx950 kernels for the STRING-MAJOR buffers (records [B][pitch][D], masked [B][pitch]).
//
// Mapping: one LANE owns one input string; one 64-lane WAVE owns a group of
// 64 consecutive strings and walks them a tile of witness rows at a time.
//   * the fused (state,byte) table of every def lives in LDS (hrx_lane.h entry format);
//     the state walk of lib.rs:804-823 is one dependent v_and_or + ds_read_b32 per row;
//   * substr-id / start / end tagging (lib.rs:825-888) rides in the low bits of the same entry;
//   * the reveal-mask scans (lib.rs:598-764) are done once per tile on per-lane position bitvectors (hrx_lane.h tile_masks);
//   * input bytes are read 16 B per lane per load; the output rows of a tile are transposed
//     through LDS so that every global store is a run of full 16-byte-per-lane lines in the
//     string-major layout a per-circuit witness-fill loop indexes.
// Pure integer/indexing work: no MFMA, HBM-bound by construction (1 B read, 4*D+2 B written per row).
#include <hip/hip_runtime.h>

#include "hrx_device.h"

namespace hrx {

// Walk one 64-row tile of this lane's string: rows t0 .. t0+63.
//   FULL: every lane of the wave has t0+64 < n, so no row needs padding treatment.
//   rem  = n - t0 (rows p >= rem are padding: lib.rs:404-418), mrem = M - 1 - t0 (end_enable of row M-1 is
//   never assigned: lib.rs:501).
// Writes the tile's compact records to this lane's LDS staging row and returns the tile bitvectors.
//
// The only serial dependency is  E_p = table[(E_{p-1} & ~0x3ff) | 4*c_p]  (one v_and_or + one ds_read_b32 per
// row and def).  The loop is software-pipelined by one row: after the lookup of row p is ISSUED, the record and
// the three bitvector bits of row p-1 are produced in the shadow of its LDS latency; the sched_barrier pins
// that order (left alone, the compiler finishes the whole chain first and keeps 64 entries live).
// linear staging row of the one-wave kernel: chunk c of this lane at my_rec + 16*c
struct LinearChunks {
    uint32_t my_rec;
    __device__ __forceinline__ uint32_t operator()(int c) const { return my_rec + 16u * (uint32_t)c; }
};
// ring slot of the walker/storer kernel: 8 chunks per lane, XOR-swizzled (addresses precomputed per tile)
struct SwizzledChunks {
    uint32_t addr[8];
    __device__ __forceinline__ uint32_t operator()(int c) const { return addr[c]; }
};

template <int D, bool FULL, int T = 64, class Chunks = LinearChunks, int NQ = T / 16, bool GTAB = false>
__device__ __forceinline__ TileBits walk_tile(LaneRegs<D> &L, const uint4 (&cq)[NQ], const WitnessArgs &a,
                                              const Chunks &chunk, int rem, int mrem, uint32_t t0) {
    uint32_t st[2] = {0, 0}, en1[2] = {0, 0}, ch[2] = {0, 0};
    uint32_t rbuf[4];
    uint32_t cw[T / 4];
#pragma unroll
    for (int i = 0; i < T / 16; ++i) { cw[4 * i] = cq[i].x; cw[4 * i + 1] = cq[i].y; cw[4 * i + 2] = cq[i].z; cw[4 * i + 3] = cq[i].w; }
    uint32_t e1[D], e2[D], raw[D];  // e1 = E_{p-1} (carries the state of row p), e2 = E_{p-2}
#pragma unroll
    for (int d = 0; d < D; ++d) e1[d] = e2[d] = L.e[d];

    // record + bits of row p from es = E_{p-1} (its state) and et = E_p (its tag)
    auto post = [&](const int p, const uint32_t (&es)[D], const uint32_t (&et)[D]) {
        uint32_t sid = 0, stn = 0, enn = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const uint32_t state = (es[d] >> kNextShift) - (d ? a.dc[d].row_base : 0u);
            uint32_t tag = et[d] & kTagMask;
            if (!FULL) {
                if (p >= mrem) tag &= ~kTagEnd;
            }
            const int slot = (p * D + d) & 3;
            rbuf[slot] = state | (tag << 16);
            if (slot == 3)
                *(__attribute__((address_space(3))) v4u32 *)(uintptr_t)chunk((p * D + d) >> 2) = v4u32{rbuf[0], rbuf[1], rbuf[2], rbuf[3]};
            if (!FULL) L.mx[d] = max(L.mx[d], et[d]);  // FULL tiles: the dead row is absorbing, the live entry after the tile tells
            sid += tag & 0xffu;
            stn += (tag >> 8) & 1u;
            enn += (tag >> 9) & 1u;
        }
        if (D > 1) {
            if (stn > 1) L.ov_row = min(L.ov_row, t0 + (uint32_t)p);
            if (enn > 1) L.ov_row = min(L.ov_row, t0 + (uint32_t)p + 1u);
        }
        st[p >> 5] |= (stn ? 1u : 0u) << (p & 31);
        en1[p >> 5] |= (enn ? 1u : 0u) << (p & 31);
        ch[p >> 5] |= (sid != L.sid_prev ? 1u : 0u) << (p & 31);
        L.sid_prev = sid;
    };

#pragma unroll
    for (int q = 0; q < T / 4; ++q) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = q * 4 + k;
            const uint32_t c4 = ((cw[q] >> (8 * k)) & 0xffu) << 2;
#pragma unroll
            for (int d = 0; d < D; ++d) raw[d] = table_at<GTAB>(a, (e1[d] & ~kTagMask) | c4);  // delta(state, byte): lib.rs:810
            if (p > 0) {
                post(p - 1, e2, e1);
                // pin the row's results here (zero instructions): IR-level sinking would otherwise move them to the tile end
                asm volatile("" : "+v"(st[(p - 1) >> 5]), "+v"(en1[(p - 1) >> 5]), "+v"(ch[(p - 1) >> 5]), "+v"(L.sid_prev));
                if (!FULL) {
#pragma unroll
                    for (int d = 0; d < D; ++d) asm volatile("" : "+v"(L.mx[d]));
                }
                if (D > 1) asm volatile("" : "+v"(L.ov_row));  // or the 64 per-row flag counts stay live until the tile end
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = 0; d < D; ++d) {
                e2[d] = e1[d];
                e1[d] = (FULL || p < rem) ? raw[d] : a.dc[d].dummy_entry;  // rows >= n: lib.rs:404-418
            }
        }
    }
    post(T - 1, e2, e1);
#pragma unroll
    for (int d = 0; d < D; ++d) {
        L.e[d] = e1[d];
        L.mx[d] = max(L.mx[d], e1[d]);
    }
    TileBits tb;
    tb.st = (uint64_t)st[0] | ((uint64_t)st[1] << 32);
    tb.en1 = (uint64_t)en1[0] | ((uint64_t)en1[1] << 32);
    tb.ch = (uint64_t)ch[0] | ((uint64_t)ch[1] << 32);
    return tb;
}

// BYTE table walk (hrx_lane.h, hrx_walk_pm.h walk_tile_pm_byte): a DFA of up to 256 states whose 4-byte table does not fit LDS (cfg 5)
// under the walker/storer kernel.  Chain: ds_read_u8 next[state << 8 | byte]; off the chain, one row later, the pair's slot of the
// perfect-hash tag table; the record and the three bitvector bits of row p - 2 are built in the shadow of both.  L.e[0] is the STATE
// here (not a table entry), L.mx[0] the largest state reached by a real transition (the dead row is the highest).
typedef __attribute__((address_space(3))) const uint8_t lds_cbyte;
__device__ __forceinline__ uint32_t lds_byte(uint32_t off) { return *(lds_cbyte *)(uintptr_t)off; }
typedef __attribute__((address_space(3))) const uint16_t lds_chalf;
__device__ __forceinline__ uint32_t lds_half(uint32_t off) { return *(lds_chalf *)(uintptr_t)off; }

template <bool FULL, int T, class Chunks>
__device__ __forceinline__ TileBits walk_tile_byte(LaneRegs<1> &L, const uint4 (&cq)[T / 16], const WitnessArgs &a, const Chunks &chunk, int rem, int mrem) {
    uint32_t st[2] = {0, 0}, en1[2] = {0, 0}, ch[2] = {0, 0};
    uint32_t rbuf[4];
    uint32_t cw[T / 4];
#pragma unroll
    for (int i = 0; i < T / 16; ++i) { cw[4 * i] = cq[i].x; cw[4 * i + 1] = cq[i].y; cw[4 * i + 2] = cq[i].z; cw[4 * i + 3] = cq[i].w; }
    const uint32_t A2 = a.byte_mul_a4 >> 1, B2 = a.byte_mul_b4 >> 1, ptab = a.byte16_ptab_off, smask = a.byte_slot_mask4 >> 1;   // this kernel stages the 2-byte slots (hrx_defs.hpp ByteTable::ptab16_off)
    uint32_t cur = L.e[0];                  // state at the row whose chain lookup is issued next
    uint32_t s1 = 0, n1 = 0;                // row p - 1: its state and its next state
    uint32_t k2 = 0, pe2 = 0;               // row p - 2: its pair (state << 8 | next) and its pair slot

    auto post = [&](const int p, const uint32_t key, const uint32_t pe) {
        uint32_t state = key >> 8;
        uint32_t tag = ((pe ^ key) & 0xffu) == 0u ? ((pe >> 8) & 0x3fu) | (pe >> 14) << 8 : 0u;   // the slot's key is this pair's next state: (state, next) is tagged
        if (!FULL) {
            if (p >= rem) tag = 0;                                   // padding rows: their lookups ran on stand-in states (lib.rs:404-418)
            if (p > rem) state = a.dc[0].dummy_state;                // lib.rs:413
            if (p >= mrem) tag &= ~kTagEnd;
        }
        rbuf[p & 3] = state | (tag << 16);
        if ((p & 3) == 3) *(__attribute__((address_space(3))) v4u32 *)(uintptr_t)chunk(p >> 2) = v4u32{rbuf[0], rbuf[1], rbuf[2], rbuf[3]};
        const uint32_t sid = tag & 0xffu;
        st[p >> 5] |= ((tag >> 8) & 1u) << (p & 31);
        en1[p >> 5] |= ((tag >> 9) & 1u) << (p & 31);
        ch[p >> 5] |= (sid != L.sid_prev ? 1u : 0u) << (p & 31);
        L.sid_prev = sid;
    };
#pragma unroll
    for (int p = 0; p < T + 2; ++p) {
        uint32_t raw_n = 0, raw_pe = 0;
        if (p < T) {
            const uint32_t c = (cw[p >> 2] >> (8 * (p & 3))) & 0xffu;
            raw_n = lds_byte((cur << 8) | c);                        // delta(state, byte): lib.rs:810
        }
        if (p >= 1 && p < T + 1) raw_pe = lds_half(((__umul24(s1, A2) + __umul24(n1, B2)) & smask) | ptab);
        if (p >= 2) {
            post(p - 2, k2, pe2);
            asm volatile("" : "+v"(st[(p - 2) >> 5]), "+v"(en1[(p - 2) >> 5]), "+v"(ch[(p - 2) >> 5]), "+v"(L.sid_prev));
        }
        __builtin_amdgcn_sched_barrier(0);
        k2 = (s1 << 8) | n1; pe2 = raw_pe;
        if (p < T) {
            const uint32_t nxt = (FULL || p < rem) ? raw_n : 0u;    // rows >= n: any valid row; post() writes the dummy state and no tag
            L.mx[0] = max(L.mx[0], nxt);                              // reaching the dead row = an undefined transition (lib.rs:817)
            asm volatile("" : "+v"(L.mx[0]));
            s1 = cur; n1 = nxt;
            cur = nxt;
        }
    }
    L.e[0] = cur;
    TileBits tb;
    tb.st = (uint64_t)st[0] | ((uint64_t)st[1] << 32);
    tb.en1 = (uint64_t)en1[0] | ((uint64_t)en1[1] << 32);
    tb.ch = (uint64_t)ch[0] | ((uint64_t)ch[1] << 32);
    return tb;
}

// 64 bytes of this lane's string, 16 B per load.  The loads are unconditional (no exec-masked merge that would
// force an early s_waitcnt): chunks that start at or beyond byte n are redirected to the string's last valid
// chunk (`last`), so nothing outside [0, stride) of the lane's own string is ever read; bytes >= n are never
// trusted (walk_tile's `live`, tile_masks' `valid`).
__device__ __forceinline__ void load_chars(uint4 (&q)[4], const uint8_t *cptr, uint32_t t0, uint32_t last) {
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = *reinterpret_cast<const uint4 *>(cptr + min(t0 + 16u * i, last));
}

// Make the compiler wait for a prefetched tile HERE (its s_waitcnt vmcnt(0) then also covers the previous tile's
// stores, issued a whole walk ago and long since drained) instead of right behind the next store burst.
template <int N>
__device__ __forceinline__ void settle_n(uint4 (&q)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(q[i].x), "+v"(q[i].y), "+v"(q[i].z), "+v"(q[i].w)::"memory");
}
__device__ __forceinline__ void settle(uint4 (&q)[4]) { settle_n(q); }

// 8 masked rows (16 B) of staged string js, rows 8w..8w+7 of the tile: masked_char | masked_substr_id << 8 (lib.rs:752-761)
template <int D>
__device__ __forceinline__ uint4 masked_chunk(uint32_t js, uint32_t w, uint32_t rec_base, uint32_t chr_base, uint32_t mb_base) {
    constexpr uint32_t RSB = 256u * D + 16u, CSB = 80u;
    const uint32_t mbyte = smem[mb_base + js * 8u + w];
    if (!mbyte) return make_uint4(0, 0, 0, 0);
    const uint2 cc = *reinterpret_cast<const uint2 *>(smem + chr_base + js * CSB + w * 8u);
    uint32_t o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint32_t sid = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) sid += (lds_u32(rec_base + js * RSB + ((w * 8u + i) * D + d) * 4u) >> 16) & 0xffu;
        const uint32_t c = ((i < 4 ? cc.x : cc.y) >> (8 * (i & 3))) & 0xffu;
        o[i] = ((mbyte >> i) & 1u) ? (c | (sid << 8)) : 0u;
    }
    return make_uint4(o[0] | (o[1] << 16), o[2] | (o[3] << 16), o[4] | (o[5] << 16), o[6] | (o[7] << 16));
}

// D: number of RegexDefs.  ALIGNED: M % 8 == 0, so every string-tile of records and masked rows starts on a
// 16-byte boundary and the store phase moves 16 B per lane.
template <int D, bool ALIGNED, bool GTAB>
__global__ __launch_bounds__(512) void witness_kernel(const WitnessArgs a) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // uniform by construction
    const uint32_t waves = blockDim.x >> 6;
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();

    // ---- stage the fused tables of all defs into LDS (offset 0) ----
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.table_image);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        if (!GTAB)
            for (uint32_t i = threadIdx.x; i < a.table_bytes / 16u; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const uint32_t lds_tab = GTAB ? 0u : a.table_bytes;  // LDS bytes the table occupies

    constexpr uint32_t RSB = 256u * D + 16u;  // bytes per staged string-tile of records (+16 B: bank spread)
    constexpr uint32_t CSB = 80u;             // bytes per staged string-tile of chars
    // A wave owns a group of GS strings (lanes >= GS idle in the walk, all 64 lanes move data in the store phase).
    // GS = 32 when the batch is too small to give every SIMD two waves of 64: with a single wave per SIMD nothing
    // can walk while that wave sits in its store burst behind HBM back-pressure (NOTES_MEASUREMENTS.md §4).
    const uint32_t GS = a.gs;
    const uint32_t rec_base = lds_tab + wave * (uint32_t)wave_stage_bytes(D, GS);
    const uint32_t chr_base = rec_base + (GS + 1u) * RSB;  // row GS of each area: scratch for the idle lanes
    const uint32_t mb_base = chr_base + (GS + 1u) * CSB;
    const uint32_t sl = lane < GS ? lane : GS;             // this lane's staging row
    const uint32_t my_rec = rec_base + sl * RSB;
    const uint32_t M = a.M;
    const uint32_t ntiles = (M + 63u) >> 6;

    for (uint32_t g = blockIdx.x * waves + wave; g < a.n_groups; g += gridDim.x * waves) {
        const uint32_t b0 = g * GS;
        const uint32_t b = b0 + lane;
        const bool active = lane < GS && b < a.B;
        const uint32_t n_raw = active ? a.lens[b] : M;
        const bool badlen = n_raw > M;
        const uint32_t n = badlen ? M : n_raw;
        const uint32_t min_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_min_u32(n));
        const uint8_t *cptr = a.chars + (size_t)(active ? b : a.B - 1u) * a.stride;
        const uint32_t last_chunk = n ? ((n - 1u) & ~15u) : 0u;

        LaneRegs<D> L;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            L.e[d] = a.dc[d].first_entry;  // states[d][0] = first_state_val: lib.rs:807
            L.mx[d] = 0;
        }
        L.sid_prev = 0;
        L.ov_row = 0xffffffffu;
        MaskCarry mc = {0, 0, 0, 0};
        uint32_t dead = 0, accept = 0;
        uint32_t err_pos[D], err_state[D], err_char[D];
#pragma unroll
        for (int d = 0; d < D; ++d) err_pos[d] = err_state[d] = err_char[d] = 0;

        uint4 cq[4], nq[4];
        load_chars(cq, cptr, 0, last_chunk);
        if (ntiles > 1) load_chars(nq, cptr, 64u, last_chunk);
        settle(cq);  // cq is plain register data from here on: no vmcnt wait may land inside a walk

        for (uint32_t t = 0; t < ntiles; ++t) {
            const uint32_t t0 = t << 6;
            unsigned long long *stamp = a.stamps ? a.stamps + ((size_t)(blockIdx.x * waves + wave) * ntiles + t) * 4u : nullptr;
            if (stamp && lane == 0) stamp[0] = __builtin_amdgcn_s_memtime();
            // ---------------- walk + tag: lib.rs:804-888 ----------------
            TileBits tb;
            const bool full = (t0 + 64u < min_n);
            if (full)
                tb = walk_tile<D, true, 64, LinearChunks, 4, GTAB>(L, cq, a, LinearChunks{my_rec}, 0, 0, t0);
            else
                tb = walk_tile<D, false, 64, LinearChunks, 4, GTAB>(L, cq, a, LinearChunks{my_rec}, (int)n - (int)t0, (int)M - 1 - (int)t0, t0);

            if (stamp && lane == 0) stamp[1] = __builtin_amdgcn_s_memtime();
            bool chars_staged = false;
            auto stage_chars = [&]() {
                if (!chars_staged) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        *reinterpret_cast<uint4 *>(smem + chr_base + sl * CSB + 16u * i) = cq[i];
                    chars_staged = true;
                }
            };

            // ---------------- undefined transition (lib.rs:817): rare slow path ----------------
            uint32_t newly = 0;
#pragma unroll
            for (int d = 0; d < D; ++d)
                if (!((dead >> d) & 1u) && L.mx[d] >= a.dc[d].dead_entry) newly |= 1u << d;
            if (__any(newly != 0)) {
                stage_chars();
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    if ((newly >> d) & 1u) {
                        const uint32_t dead_state = a.dc[d].n_rows - 1u;
                        for (uint32_t p = 0; p < 64u; ++p) {
                            const uint32_t s_p = lds_u32(my_rec + (p * D + d) * 4u) & 0xffffu;
                            const uint32_t s_n = (p < 63u) ? (lds_u32(my_rec + ((p + 1u) * D + d) * 4u) & 0xffffu)
                                                           : ((L.e[d] >> kNextShift) - a.dc[d].row_base);
                            if (s_n == dead_state && s_p != dead_state) {
                                err_pos[d] = t0 + p;
                                err_state[d] = s_p;
                                err_char[d] = smem[chr_base + sl * CSB + p];
                                break;
                            }
                        }
                        dead |= 1u << d;
                    }
                }
            }

            // ---------------- accept state: the state at row n (lib.rs:437-457) ----------------
            if (!full) {
                if (n >= t0 && n < t0 + 64u) {
                    accept = 0;
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        const uint32_t s_n = lds_u32(my_rec + ((n - t0) * D + d) * 4u) & 0xffffu;
                        accept |= (s_n == a.dc[d].accepted_state ? 1u : 0u) << d;
                    }
                } else if (n == t0 + 64u && t + 1 == ntiles) {  // n == M: row n does not exist, s[n] is the live state
                    accept = 0;
#pragma unroll
                    for (int d = 0; d < D; ++d)
                        accept |= (((L.e[d] >> kNextShift) - a.dc[d].row_base) == a.dc[d].accepted_state ? 1u : 0u) << d;
                }
            }

            // ---------------- reveal masks: lib.rs:598-764 ----------------
            TileMasks tm = tile_masks(tb, mc, t0, tile_is_exact(t0, n, M), rows_below(t0, n));
            if (!active) { tm.mask = 0; tm.fix = 0; }
            *reinterpret_cast<uint64_t *>(smem + mb_base + sl * 8u) = tm.mask;
            const bool any_mask = __any(tm.mask != 0);
            if (any_mask) stage_chars();

            // rotate the char tiles BEFORE the store burst: the wait for tile t+1's bytes lands here, where every
            // older vector-memory op (tile t-1's stores, the loads themselves) finished long ago; tile t+2's loads
            // are issued ahead of this tile's stores so that they never queue behind them.
            if (t + 1 < ntiles) {
                settle(nq);
#pragma unroll
                for (int i = 0; i < 4; ++i) cq[i] = nq[i];
            }
            if (t + 2 < ntiles && !(a.debug & kDbgInputFromL2)) load_chars(nq, cptr, t0 + 128u, last_chunk);

            // an earlier optimistic end_mask = 1 turned out wrong: zero those masked rows (rare)
            uint64_t fixm = __ballot(tm.fix != 0);
            if (fixm) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                while (fixm) {
                    const int j = __ffsll((unsigned long long)fixm) - 1;
                    fixm &= fixm - 1;
                    const uint32_t fs = (uint32_t)__builtin_amdgcn_readlane((int)tm.fix_start, j);
                    uint16_t *mrow = a.masked + (size_t)(b0 + j) * a.msk_pitch;
                    for (uint32_t r = (fs & ~63u) + lane; r < t0; r += 64u)
                        if (r >= fs) mrow[r] = 0;
                }
            }

            if (stamp && lane == 0) stamp[2] = __builtin_amdgcn_s_memtime();
            // ---------------- store phase: LDS-transposed, coalesced ----------------
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const bool whole = (b0 + GS <= a.B) && (t0 + 64u <= M);  // wave-uniform: no partial string group / tile
            if (ALIGNED && whole && (64u % (16u * D)) == 0u) {
                // Fast path.  A string-tile of records is 256*D contiguous bytes = CPS chunks of 16 B; one
                // wave-instruction moves SPI whole string-tiles (1 KiB).  All LDS reads of a batch are issued
                // before the first store, addresses advance by a uniform step: no waits, no branches.
                constexpr uint32_t CPS = 16u * D, SPI = 64u / CPS;
                const uint32_t js0 = lane / CPS, w = lane % CPS;
                uint32_t lds_a = rec_base + js0 * RSB + w * 16u;
                unsigned char *gp = reinterpret_cast<unsigned char *>(a.records + ((size_t)(b0 + js0) * a.rec_pitch + t0) * D + w * 4u);
                const size_t gstep = (size_t)SPI * a.rec_pitch * D * 4u;
                if (!(a.debug & kDbgSkipRecords)) {
                    const uint32_t nit = CPS * GS / 64u;  // wave-instructions that cover the group's GS string-tiles
                    for (uint32_t it0 = 0; it0 < nit; it0 += 8u) {
                        uint4 v[8];
#pragma unroll
                        for (uint32_t i = 0; i < 8u; ++i)
                            if (it0 + i < nit) v[i] = lds_u128(lds_a + (it0 + i) * (SPI * RSB));
#pragma unroll
                        for (uint32_t i = 0; i < 8u; ++i) {
                            if (it0 + i < nit) { if (a.debug & kDbgNoNtStores) *reinterpret_cast<uint4 *>(gp) = v[i]; else store16_nt(gp, v[i]); }   // full lines, written once
                            gp += gstep;
                        }
                    }
                }
                // masked rows: a string-tile is 128 contiguous bytes = 8 chunks of 16 B (8 rows each)
                const uint32_t mj0 = lane >> 3, mw = lane & 7u;
                unsigned char *mp = reinterpret_cast<unsigned char *>(a.masked + (size_t)(b0 + mj0) * a.msk_pitch + t0 + mw * 8u);
                const size_t mstep = (size_t)8u * a.msk_pitch * 2u;
                if (!(a.debug & kDbgSkipMasked)) {
                    if (!any_mask) {
                        for (uint32_t it = 0; it < GS / 8u; ++it) {
                            store16_nt(mp, make_uint4(0, 0, 0, 0));
                            mp += mstep;
                        }
                    } else {
                        for (uint32_t it = 0; it < GS / 8u; ++it) {
                            store16_nt(mp, masked_chunk<D>(it * 8u + mj0, mw, rec_base, chr_base, mb_base));
                            mp += mstep;
                        }
                    }
                }
            } else if (ALIGNED) {
                // partial string group or partial tile (or D = 3): same mapping, predicated per chunk
                constexpr uint32_t CPS = 16u * D;
                const uint32_t lim = (M - t0 >= 64u ? 64u : M - t0) * D / 4u;  // valid chunks per string-tile
#pragma unroll 4
                for (uint32_t it = 0; it < CPS; ++it) {
                    const uint32_t chunk = it * 64u + lane;
                    const uint32_t js = chunk / CPS, w = chunk % CPS;
                    if (js < GS && b0 + js < a.B && w < lim && !(a.debug & kDbgSkipRecords)) {
                        const uint4 v = lds_u128(rec_base + js * RSB + w * 16u);
                        uint32_t *dst = a.records + ((size_t)(b0 + js) * a.rec_pitch + t0) * D + w * 4u;
                        *reinterpret_cast<uint4 *>(dst) = v;
                    }
                }
                const uint32_t mlim = (M - t0 >= 64u ? 64u : M - t0) / 8u;
#pragma unroll 2
                for (uint32_t it = 0; it < 8u; ++it) {
                    const uint32_t js = it * 8u + (lane >> 3), w = lane & 7u;
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (any_mask && js < GS) v = masked_chunk<D>(js, w, rec_base, chr_base, mb_base);
                    if (js < GS && b0 + js < a.B && w < mlim && !(a.debug & kDbgSkipMasked))
                        *reinterpret_cast<uint4 *>(a.masked + (size_t)(b0 + js) * a.msk_pitch + t0 + w * 8u) = v;
                }
            } else {
                // generic M: one dword / one u16 per lane, still contiguous per string
                const uint32_t rows = (M - t0 >= 64u ? 64u : M - t0);
                for (uint32_t js = 0; js < GS && b0 + js < a.B; ++js) {
#pragma unroll
                    for (int dd = 0; dd < D; ++dd) {
                        const uint32_t i = dd * 64u + lane;
                        if (i < rows * D)
                            a.records[((size_t)(b0 + js) * a.rec_pitch + t0) * D + i] = lds_u32(rec_base + js * RSB + i * 4u);
                    }
                    if (lane < rows) {
                        uint32_t o = 0;
                        if (any_mask) {
                            const uint64_t mbits = *reinterpret_cast<const uint64_t *>(smem + mb_base + js * 8u);
                            if ((mbits >> lane) & 1ull) {
                                uint32_t sid = 0;
#pragma unroll
                                for (int d = 0; d < D; ++d) sid += (lds_u32(rec_base + js * RSB + (lane * D + d) * 4u) >> 16) & 0xffu;
                                o = smem[chr_base + js * CSB + lane] | (sid << 8);
                            }
                        }
                        a.masked[(size_t)(b0 + js) * a.msk_pitch + t0 + lane] = (uint16_t)o;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (stamp && lane == 0) stamp[3] = __builtin_amdgcn_s_memtime();

        }

        // ---------------- per-string status ----------------
        if (active) {
            uint64_t sw;
            if (badlen) {
                sw = kStatusBadLength;
            } else if (dead) {
                sw = 0;
#pragma unroll
                for (int d = D - 1; d >= 0; --d)  // lowest def wins: the reference walks defs in order (lib.rs:806)
                    if ((dead >> d) & 1u) sw = status_invalid((uint32_t)d, err_pos[d], err_state[d], err_char[d]);
            } else if (D > 1 && L.ov_row != 0xffffffffu) {
                sw = status_overlap(L.ov_row);
            } else {
                sw = status_ok(accept);
            }
            a.status[b] = sw;
        }
    }
}

// =============================================================================================
// Walker / storer kernel (the production path for D = 1 and D = 2, M % 8 == 0).
//
// Measured on MI355X (profiles/, NOTES_MEASUREMENTS.md §4): with one wave doing everything, each tile's store burst
// sits behind HBM back-pressure for 5-12k cycles (868 when the chip is idle) while the wave cannot walk, and
// the walk leaves HBM idle — compute time and memory time ADD.  Here each CU runs 4 pairs of waves:
//   walker  (wave i)   : everything that is per-string and sequential — the dependent table walk, record
//                        build, flag bitvectors, reveal-mask scans, accept/dead/overlap bookkeeping
//                        (lib.rs:804-888, 598-764) — T rows at a time into an LDS ring slot.  It issues no
//                        global stores in the loop, so only a full ring can stall the chain.
//   storer  (wave i+4) : a pure mover, LDS slot -> coalesced global stores (+ the rare masked-row assembly and
//                        fix-ups).  It is the wave that absorbs HBM back-pressure.
// A pair shares a ring of NSLOTS slots; `prod`/`cons` tile counters in LDS hand slots over (LDS is one
// coherent, in-order memory per CU; the release is s_waitcnt lgkmcnt(0) before the counter store).
// Slot = 64 strings x 128 B of records (T rows x D defs x 4 B; 16-byte chunks XOR-swizzled with the lane so
// that the walker's ds_write_b128 and the storer's transposed ds_read_b128 are conflict-free without padding)
// + per string {reveal mask word, fix-up start} + the tile's input bytes (the storer needs them for the masked
// rows and must not issue global loads: a load would queue behind its own stores).
// =============================================================================================
template <int D, int T>
struct SplitGeom {
    static_assert(T * D * 4 == 128, "a string-tile of records is 128 bytes");
    static constexpr uint32_t kSlotRec = 64u * 128u;
    static constexpr uint32_t kSlotHdr = kSlotRec;               // {mask, fix_start} per string
    static constexpr uint32_t kSlotChr = kSlotRec + 64u * 8u;    // the tile's T input bytes per string (written only when some mask bit is set)
    static constexpr uint32_t kSlotBytes = kSlotChr + 64u * T;
    static constexpr uint32_t kPairFixed = 16u;                  // prod / cons
};
// BYTE: the 1-byte next-state table + perfect-hash pair tags (hrx_lane.h) instead of the 4-byte fused table: one def of up to 256 states
// whose 4-byte table does not fit LDS (cfg 5) — its string-major outputs used to take the position-major kernel + a transpose launch.
template <int D, int T, bool BYTE = false>
__global__ __launch_bounds__(512) void witness_split_kernel(const WitnessArgs a, const uint32_t nslots) {
    static_assert(!BYTE || D == 1, "the BYTE table serves one def");
    using G = SplitGeom<D, T>;
    const uint32_t tab_bytes = BYTE ? a.byte16_bytes : a.table_bytes;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t pairs = blockDim.x >> 7;  // walker waves 0..pairs-1, storer waves pairs..2*pairs-1
    const bool is_walker = wave < pairs;
    const uint32_t pair = is_walker ? wave : wave - pairs;
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();

    const uint32_t pair_bytes = nslots * G::kSlotBytes + G::kPairFixed;
    const uint32_t ring_base = tab_bytes + pair * pair_bytes;
    const uint32_t prod_off = ring_base + nslots * G::kSlotBytes, cons_off = prod_off + 4u;
    const uint32_t scratch_off = tab_bytes + pairs * pair_bytes + pair * 256u;  // 256 B per storer: LDS-DMA sink
    {
        const uint4 *src = BYTE ? reinterpret_cast<const uint4 *>(a.byte_image) : reinterpret_cast<const uint4 *>(a.table_image);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        if (BYTE) {   // next-state bytes, then the 2-byte pair slots that follow the position-major kernel's image in the device blob
            for (uint32_t i = threadIdx.x; i < a.byte_rows_bytes / 16u; i += blockDim.x) dst[i] = src[i];
            const uint4 *src16 = reinterpret_cast<const uint4 *>(a.byte_image + a.byte_bytes);
            uint4 *dst16 = reinterpret_cast<uint4 *>(smem + a.byte16_ptab_off);
            for (uint32_t i = threadIdx.x; i < (tab_bytes - a.byte16_ptab_off) / 16u; i += blockDim.x) dst16[i] = src16[i];
        } else {
            for (uint32_t i = threadIdx.x; i < tab_bytes / 16u; i += blockDim.x) dst[i] = src[i];
        }
        if (is_walker && lane == 0) { lds_store_u32(prod_off, 0); lds_store_u32(cons_off, 0); }
    }
    __syncthreads();

    const uint32_t M = a.M;
    const uint32_t ntiles = (M + T - 1u) / T;
    const uint32_t l7 = lane & 7u;
    uint32_t seq = 0, cons_seen = 0;  // tiles handed over by this pair so far

    for (uint32_t g = blockIdx.x * pairs + pair; g < a.n_groups; g += gridDim.x * pairs) {
        const uint32_t b0 = g * 64u;
        const uint32_t b = b0 + lane;
        const bool active = b < a.B;

        if (is_walker) {
            // ================================ walker ================================
            const uint32_t n_raw = active ? a.lens[b] : M;
            const bool badlen = n_raw > M;
            const uint32_t n = badlen ? M : n_raw;
            const uint32_t min_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_min_u32(n));
            const uint8_t *cptr = a.chars + (size_t)(active ? b : a.B - 1u) * a.stride;
            const uint32_t last_chunk = n ? ((n - 1u) & ~15u) : 0u;
            LaneRegs<D> L;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                L.e[d] = BYTE ? a.dc[d].first_state : a.dc[d].first_entry;  // states[d][0] = first_state_val: lib.rs:807
                L.mx[d] = 0;
            }
            L.sid_prev = 0;
            L.ov_row = 0xffffffffu;
            MaskCarry mc = {0, 0, 0, 0};
            uint32_t dead = 0, accept = 0;
            uint32_t err_pos[D], err_state[D], err_char[D];
#pragma unroll
            for (int d = 0; d < D; ++d) err_pos[d] = err_state[d] = err_char[d] = 0;

            // Input bytes: 16 B per lane per load, fetched kSuper tiles at a time into `pen` while the walker consumes
            // `act`.  While the storers saturate the HBM write path a read takes several microseconds (measured: a
            // 2-tile distance left the walker stalled ~0.85 us per tile), so a batch is requested kSuper tiles before
            // its first use; the walker has no stores, and when it waits for `pen` nothing else is outstanding, so
            // the compiler's s_waitcnt vmcnt(0) there is exact and no wait ever lands inside a walk.
            constexpr int CPT = T / 16;  // 16-byte chunks per tile
            constexpr int kSuper = 4;    // tiles per batch
            uint4 act[kSuper * CPT], pen[kSuper * CPT];
            auto load_batch = [&](uint4 (&dst)[kSuper * CPT], uint32_t first_row) {
#pragma unroll
                for (int i = 0; i < kSuper * CPT; ++i)
                    dst[i] = *reinterpret_cast<const uint4 *>(cptr + min(first_row + 16u * i, last_chunk));
            };
            load_batch(act, 0);
            load_batch(pen, kSuper * T);
            settle_n(act);
            for (uint32_t t = 0; t < ntiles; ++t, ++seq) {
                const uint32_t t0 = t * T;
                const uint32_t slot = ring_base + (seq % nslots) * G::kSlotBytes;
                SwizzledChunks chunks;
#pragma unroll
                for (uint32_t c = 0; c < 8u; ++c) chunks.addr[c] = slot + lane * 128u + ((c ^ l7) << 4);
                unsigned long long *stamp = a.stamps ? a.stamps + ((size_t)(blockIdx.x * pairs + pair) * ntiles + t) * 8u : nullptr;
                if (stamp && lane == 0) stamp[0] = __builtin_amdgcn_s_memtime();
                if (seq >= nslots) ring_wait_seen(cons_off, seq - nslots + 1u, cons_seen);  // the storer has drained this slot
                if (stamp && lane == 0) stamp[1] = __builtin_amdgcn_s_memtime();

                // ---------------- walk + tag: lib.rs:804-888 ----------------
                TileBits tb = {0, 0, 0};
                const bool full = (t0 + T < min_n);
                if (a.debug & kDbgSplitNoWalk) {
                    // profiling only: no walk, the storer moves whatever the slot holds
                } else if constexpr (BYTE) {
                    const uint4 (&cqt)[CPT] = reinterpret_cast<const uint4 (&)[CPT]>(act);
                    if (full) tb = walk_tile_byte<true, T>(L, cqt, a, chunks, 0, 0);
                    else tb = walk_tile_byte<false, T>(L, cqt, a, chunks, (int)n - (int)t0, (int)M - 1 - (int)t0);
                } else if (full)
                    tb = walk_tile<D, true, T>(L, act, a, chunks, 0, 0, t0);
                else
                    tb = walk_tile<D, false, T>(L, act, a, chunks, (int)n - (int)t0, (int)M - 1 - (int)t0, t0);

                auto rec_addr = [&](uint32_t i) { return slot + lane * 128u + (((i >> 2) ^ l7) << 4) + (i & 3u) * 4u; };
                // ---------------- undefined transition (lib.rs:817): rare slow path ----------------
                uint32_t newly = 0;
#pragma unroll
                for (int d = 0; d < D; ++d)
                    if (!((dead >> d) & 1u) && (BYTE ? L.mx[d] >= a.byte_dead : L.mx[d] >= a.dc[d].dead_entry)) newly |= 1u << d;
                if (__any(newly != 0)) {
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        if ((newly >> d) & 1u) {
                            const uint32_t dead_state = BYTE ? a.byte_dead : a.dc[d].n_rows - 1u;
                            for (uint32_t p = 0; p < (uint32_t)T; ++p) {
                                const uint32_t s_p = lds_u32(rec_addr(p * D + d)) & 0xffffu;
                                const uint32_t s_n = (p + 1u < (uint32_t)T) ? (lds_u32(rec_addr((p + 1u) * D + d)) & 0xffffu)
                                                                            : BYTE ? L.e[d] : ((L.e[d] >> kNextShift) - a.dc[d].row_base);
                                if (s_n == dead_state && s_p != dead_state) {
                                    err_pos[d] = t0 + p;
                                    err_state[d] = s_p;
                                    err_char[d] = cptr[t0 + p];
                                    break;
                                }
                            }
                            dead |= 1u << d;
                        }
                    }
                }
                // ---------------- accept state: the state at row n (lib.rs:437-457) ----------------
                if (!full) {
                    if (n >= t0 && n < t0 + T) {
                        accept = 0;
#pragma unroll
                        for (int d = 0; d < D; ++d)
                            accept |= ((lds_u32(rec_addr((n - t0) * D + d)) & 0xffffu) == a.dc[d].accepted_state ? 1u : 0u) << d;
                    } else if (n == t0 + T && t + 1 == ntiles) {  // n == M: row n does not exist, s[n] is the live state
                        accept = 0;
#pragma unroll
                        for (int d = 0; d < D; ++d)
                            accept |= ((BYTE ? L.e[d] : (L.e[d] >> kNextShift) - a.dc[d].row_base) == a.dc[d].accepted_state ? 1u : 0u) << d;
                    }
                }
                // ---------------- reveal masks: lib.rs:598-764 ----------------
                TileMasks tm = tile_masks<T>(tb, mc, t0, tile_is_exact(t0, n, M, T), rows_below(t0, n));
                if (!active) { tm.mask = 0; tm.fix = 0; }
                if (a.debug & kDbgSkipFixups) tm.fix = 0;  // profiling only
                *(__attribute__((address_space(3))) v2u32 *)(uintptr_t)(slot + G::kSlotHdr + lane * 8u) =
                    v2u32{(uint32_t)tm.mask, tm.fix ? tm.fix_start : kNoFix};
                if (__any(tm.mask != 0)) {
#pragma unroll
                    for (int i = 0; i < CPT; ++i)
                        *(__attribute__((address_space(3))) v4u32 *)(uintptr_t)(slot + G::kSlotChr + lane * T + 16u * i) =
                            v4u32{act[i].x, act[i].y, act[i].z, act[i].w};
                }
                ring_post_lds(prod_off, seq + 1u);
                cons_seen = lds_vol_u32(cons_off);      // the next tile's look at the storer's counter (hrx_device.h ring_wait_seen)
                if (stamp && lane == 0) stamp[2] = __builtin_amdgcn_s_memtime();

                // next tile's bytes move to the front; every kSuper tiles the pending batch takes over
                if ((t + 1) % kSuper != 0) {
#pragma unroll
                    for (int i = 0; i < (kSuper - 1) * CPT; ++i) act[i] = act[i + CPT];
                } else {
#pragma unroll
                    for (int i = 0; i < kSuper * CPT; ++i) act[i] = pen[i];
                    settle_n(act);
                    if (!(a.debug & kDbgInputFromL2)) load_batch(pen, t0 + T + kSuper * T);
                }
                if (stamp && lane == 0) stamp[3] = __builtin_amdgcn_s_memtime();
            }
            // ---------------- per-string status ----------------
            if (active) {
                uint64_t sw;
                if (badlen) {
                    sw = kStatusBadLength;
                } else if (dead) {
                    sw = 0;
#pragma unroll
                    for (int d = D - 1; d >= 0; --d)  // lowest def wins: the reference walks defs in order (lib.rs:806)
                        if ((dead >> d) & 1u) sw = status_invalid((uint32_t)d, err_pos[d], err_state[d], err_char[d]);
                } else if (D > 1 && L.ov_row != 0xffffffffu) {
                    sw = status_overlap(L.ov_row);
                } else {
                    sw = status_ok(accept);
                }
                a.status[b] = sw;
            }
        } else {
            // ================================ storer ================================
            // Masked rows are 2 B each: a 32-row tile is only half a 128-byte line per string, and half-line
            // stores interleaved with the record stream cost more than everything else together (measured:
            // 138 us vs 65 us per launch).  So the masked rows of an even tile wait in registers (mk) and go out
            // with the odd tile's: 64 rows = one full line per string.
            // L2 warm-up for the walker.  Behind the saturated write path an HBM read takes ~10 us, more than the walker's
            // prefetch distance can cover, and its vmcnt is in-order, so it cannot run far-ahead loads itself.  The storer
            // never waits on vmcnt, so every 128 rows it issues ONE LDS-DMA load (no VGPR destination; 4 bytes per lane into
            // a scratch word nobody reads) that pulls each string's 128-byte line of kTouch batches ahead into L2.
            // (Not on the BYTE table — cfg 5's 4096-byte strings: there the touched lines are evicted again before the walker reads them, the input crosses
            // the HBM twice (2179 against 1879 MB per launch, profiles/r03_cfg5_sm_pmc.json) and the launch is 4 % slower with the touches than without.)
            constexpr uint32_t kBlk = 64u;
            constexpr uint32_t kTouchRows = 128u, kTouchAhead = 5u * 128u;
            const uint32_t n_s = active ? min(a.lens[b], M) : M;
            const uint32_t last_line = n_s ? ((n_s - 1u) & ~127u) : 0u;
            const uint8_t *tptr = a.chars + (size_t)(active ? b : a.B - 1u) * a.stride;
            auto touch = [&](uint32_t row) {
                if (!(a.debug & kDbgNoTouch) && !a.sm_no_touch) {
                    uint32_t saved_m0;  // M0 = LDS base of the DMA; restored, the compiler does not expect it to change
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(saved_m0)
                                 : "v"(tptr + min(row, last_line)), "s"(scratch_off)
                                 : "memory");
                }
            };
            for (uint32_t r = 2u * kTouchRows; r <= kTouchAhead; r += kTouchRows) touch(r);
            uint4 mk[8];
            const uint32_t mj0 = lane >> 3, mw = lane & 7u;  // masked block mapping: string (it*8 + mj0), rows 8*mw..8*mw+7
            for (uint32_t t = 0; t < ntiles; ++t, ++seq) {
                const uint32_t t0 = t * T;
                const uint32_t blk0 = t0 & ~(kBlk - 1u);
                const uint32_t sub = (t0 - blk0) / T;  // which T-row part of the 64-row block this tile is
                const bool blk_last = (t0 + T >= blk0 + kBlk) || (t + 1 == ntiles);
                const uint32_t slot = ring_base + (seq % nslots) * G::kSlotBytes;
                unsigned long long *stamp = a.stamps ? a.stamps + ((size_t)(blockIdx.x * pairs + pair) * ntiles + t) * 8u : nullptr;
                if (stamp && lane == 0) stamp[4] = __builtin_amdgcn_s_memtime();
                ring_wait(prod_off, seq + 1u);
                if (stamp && lane == 0) stamp[5] = __builtin_amdgcn_s_memtime();
                const v2u32 hdr = *(__attribute__((address_space(3))) const v2u32 *)(uintptr_t)(slot + G::kSlotHdr + lane * 8u);
                const bool any_mask = __any(hdr.x != 0);
                if (sub == 0) {
#pragma unroll
                    for (int it = 0; it < 8; ++it) mk[it] = make_uint4(0, 0, 0, 0);
                }
                if (__ballot(hdr.y != kNoFix)) {  // an earlier optimistic end_mask = 1 turned out wrong: zero those masked rows
                    // rows already in memory (everything below this 64-row block): one string at a time, behind the stores that wrote them
                    uint64_t memfix = __ballot(hdr.y < blk0);          // (kNoFix = 0xffffffff)
                    if (memfix) {      // (this wave's own earlier stores to these rows are performed before these: one wave's stores to one address keep their order, as in the position-major kernels' repairs)
                        while (memfix) {
                            const int j = __ffsll((unsigned long long)memfix) - 1;
                            memfix &= memfix - 1;
                            const uint32_t fs = (uint32_t)__builtin_amdgcn_readlane((int)hdr.y, j);
                            uint16_t *mrow = a.masked + (size_t)(b0 + j) * a.msk_pitch;
                            if (((a.msk_pitch & 7u) | ((uint32_t)(uintptr_t)a.masked & 15u)) == 0u) {
                                // the rows up to the next octet border two bytes at a time (lanes 0 .. 6), then 16 bytes per lane: 512 rows per store instruction (blk0 is a multiple of 64) —
                                // a random DFA's repairs reach back hundreds of rows (hrx_kernel_pm.hip does the same on its layout)
                                const uint32_t o0 = (fs + 7u) >> 3, o1 = blk0 >> 3;
                                if (fs + lane < (o0 << 3)) mrow[fs + lane] = 0;
                                for (uint32_t o = o0 + lane; o < o1; o += 64u) *reinterpret_cast<uint4 *>(mrow + (size_t)o * 8u) = make_uint4(0, 0, 0, 0);
                            } else {
                                for (uint32_t r = (fs & ~63u) + lane; r < blk0; r += 64u)
                                    if (r >= fs) mrow[r] = 0;
                            }
                        }
                    }
                    // rows of this block still held in mk (earlier tiles of the block): every lane looks at its own eight strings' fix starts —
                    // all strings at once (a loop over the fixing strings cost cfg 5, where a fifth of the string-tiles repairs, 0.07 of 0.56 ms)
                    const uint32_t r0 = blk0 + mw * 8u;  // first row of this lane's chunk
                    if (t0 > blk0 && r0 < t0) {
#pragma unroll
                        for (int it = 0; it < 8; ++it) {
                            const uint32_t fs = lds_u32(slot + G::kSlotHdr + ((uint32_t)it * 8u + mj0) * 8u + 4u);
                            if (fs < t0 && r0 + 8u > fs) {
                                const uint32_t keep = fs > r0 ? fs - r0 : 0u;  // leading rows that stay
                                uint32_t wds[4] = {mk[it].x, mk[it].y, mk[it].z, mk[it].w};
#pragma unroll
                                for (uint32_t e = 0; e < 8u; ++e)
                                    if (e >= keep) wds[e >> 1] &= (e & 1u) ? 0x0000ffffu : 0xffff0000u;
                                mk[it] = make_uint4(wds[0], wds[1], wds[2], wds[3]);
                            }
                        }
                    }
                }
                const bool whole = (b0 + 64u <= a.B) && (t0 + T <= M);
                const uint32_t rows = M > t0 ? (M - t0 >= (uint32_t)T ? (uint32_t)T : M - t0) : 0u;
                {   // records: 8 chunks of 16 B per string-tile, 8 strings per wave-instruction, 8 instructions
                    const uint32_t js0 = lane >> 3, w = lane & 7u;
                    uint4 v[8];
#pragma unroll
                    for (uint32_t it = 0; it < 8u; ++it) {
                        const uint32_t js = it * 8u + js0;
                        v[it] = lds_u128(slot + js * 128u + ((w ^ (js & 7u)) << 4));
                    }
                    unsigned char *gp = reinterpret_cast<unsigned char *>(a.records + ((size_t)(b0 + js0) * a.rec_pitch + t0) * D + w * 4u);
                    const size_t gstep = (size_t)8u * a.rec_pitch * D * 4u;
                    const uint32_t lim = rows * D / 4u;
                    // streaming stores, except every k-th 64-row block's records (plan_nt_mix: ~128 MiB of a launch's records write-back)
                    const uint32_t wb_k = a.nt_mix & 0xffu;
                    const bool rec_wb = (a.debug & kDbgNoNtStores) || (wb_k != 0u && ((t0 >> 6) % wb_k) == wb_k - 1u);
                    if (!(a.debug & kDbgSkipRecords)) {
#pragma unroll
                        for (uint32_t it = 0; it < 8u; ++it) {
                            if (whole || (b0 + it * 8u + js0 < a.B && w < lim)) { if (rec_wb) *reinterpret_cast<uint4 *>(gp) = v[it]; else store16_nt(gp, v[it]); }   // full 128-byte lines, 8 strings per instruction
                            gp += gstep;
                        }
                    }
                }
                // masked rows of this tile -> mk (lanes whose chunk lies in this tile)
                if (any_mask) {
                    constexpr uint32_t CM = T / 8u;  // 8-row chunks per tile
                    if (mw / CM == sub) {
                        const uint32_t w = mw % CM;
#pragma unroll
                        for (int it = 0; it < 8; ++it) {
                            const uint32_t js = (uint32_t)it * 8u + mj0;
                            const uint32_t mbyte = smem[slot + G::kSlotHdr + js * 8u + w];
                            if (mbyte) {
                                const uint2 cc = *reinterpret_cast<const uint2 *>(smem + slot + G::kSlotChr + js * T + w * 8u);
                                uint32_t o[8];
#pragma unroll
                                for (int i = 0; i < 8; ++i) {
                                    uint32_t sid = 0;
#pragma unroll
                                    for (int d = 0; d < D; ++d) {
                                        const uint32_t ix = (w * 8u + i) * D + d;
                                        sid += (lds_u32(slot + js * 128u + (((ix >> 2) ^ (js & 7u)) << 4) + (ix & 3u) * 4u) >> 16) & 0xffu;
                                    }
                                    const uint32_t c = ((i < 4 ? cc.x : cc.y) >> (8 * (i & 3))) & 0xffu;
                                    o[i] = ((mbyte >> i) & 1u) ? (c | (sid << 8)) : 0u;  // lib.rs:752-761
                                }
                                mk[it] = make_uint4(o[0] | (o[1] << 16), o[2] | (o[3] << 16), o[4] | (o[5] << 16), o[6] | (o[7] << 16));
                            }
                        }
                    }
                }
                ring_post_lds(cons_off, seq + 1u);  // behind every LDS read of the slot; the stores may still be in flight
                if (t0 % kTouchRows == 0) touch(t0 + kTouchAhead + kTouchRows);
                if (blk_last && !(a.debug & kDbgSkipMasked)) {
                    unsigned char *mp = reinterpret_cast<unsigned char *>(a.masked + (size_t)(b0 + mj0) * a.msk_pitch + blk0 + mw * 8u);
                    const size_t mstep = (size_t)8u * a.msk_pitch * 2u;
                    const bool blk_whole = (b0 + 64u <= a.B) && (blk0 + kBlk <= M);
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        if (blk_whole || (b0 + (uint32_t)it * 8u + mj0 < a.B && blk0 + mw * 8u < M)) { if (a.debug & kDbgNoNtStores) *reinterpret_cast<uint4 *>(mp) = mk[it]; else store16_nt(mp, mk[it]); }
                        mp += mstep;
                    }
                }
                if (stamp && lane == 0) stamp[6] = __builtin_amdgcn_s_memtime();
            }
        }
    }
}

template <int D, int T, bool BYTE = false>
static hipError_t launch_split(const WitnessArgs &a, const LaunchInfo &li, hipStream_t stream) {
    auto k = witness_split_kernel<D, T, BYTE>;
    static std::atomic<size_t> granted[64];  // per device: the attribute is set on the current device's function
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipError_t e = ensure_lds(k, granted[dev & 63], li.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(li.grid), dim3(64 * li.waves_per_wg), li.lds_bytes, stream, a, (uint32_t)li.nslots);
    return hipGetLastError();
}

template <int D, bool ALIGNED, bool GTAB>
static hipError_t launch_t(const WitnessArgs &a, const LaunchInfo &li, hipStream_t stream) {
    auto k = witness_kernel<D, ALIGNED, GTAB>;
    static std::atomic<size_t> granted[64];  // per device: the attribute is set on the current device's function
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipError_t e = ensure_lds(k, granted[dev & 63], li.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(li.grid), dim3(64 * li.waves_per_wg), li.lds_bytes, stream, a);
    return hipGetLastError();
}

hipError_t launch_witness_sm(const WitnessArgs &a, const LaunchInfo &li, hipStream_t stream) {
    if (li.split) return li.byte ? launch_split<1, 32, true>(a, li, stream) : a.D == 1 ? launch_split<1, 32>(a, li, stream) : launch_split<2, 16>(a, li, stream);
    const bool al = (a.M % 8u) == 0;
    if (li.gtab) {
        switch (a.D) {
            case 1: return al ? launch_t<1, true, true>(a, li, stream) : launch_t<1, false, true>(a, li, stream);
            case 2: return al ? launch_t<2, true, true>(a, li, stream) : launch_t<2, false, true>(a, li, stream);
            case 3: return al ? launch_t<3, true, true>(a, li, stream) : launch_t<3, false, true>(a, li, stream);
            default: return hipErrorInvalidValue;
        }
    }
    switch (a.D) {
        case 1: return al ? launch_t<1, true, false>(a, li, stream) : launch_t<1, false, false>(a, li, stream);
        case 2: return al ? launch_t<2, true, false>(a, li, stream) : launch_t<2, false, false>(a, li, stream);
        case 3: return al ? launch_t<3, true, false>(a, li, stream) : launch_t<3, false, false>(a, li, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace hrx
