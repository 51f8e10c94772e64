// hrx_kernel.hip — gfx950 (CDNA4) kernels of the batched DFA witness generator.
//
// Mapping: one LANE owns one input string; one 64-lane WAVE owns a group of
// 64 consecutive strings and walks them 64 witness rows (one tile) at a time.
//   * the fused (state,byte) table of every def lives in LDS (hrx_lane.h entry format);
//     the state walk of lib.rs:804-823 is one dependent v_and_or + ds_read_b32 per row;
//   * substr-id / start / end tagging (lib.rs:825-888) rides in the low bits of the same entry;
//   * the reveal-mask scans (lib.rs:598-764) are done once per tile on per-lane 64-bit position
//     bitvectors (hrx_lane.h tile_masks);
//   * input bytes are read 16 B per lane per load; the output rows of a tile are transposed
//     through LDS so that every global store is a run of full 16-byte-per-lane lines in the
//     string-major layout the witness-fill side consumes.
// Pure integer/indexing work: no MFMA, HBM-bound by construction (1 B read, 4*D+2 B written per row).
#include <hip/hip_runtime.h>

#include <atomic>

#include "hrx_fr.h"
#include "hrx_kernel.hpp"
#include "hrx_lane.h"

namespace hrx {

extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

// The kernels declare no static LDS, so the dynamic segment starts at LDS address 0 and a byte offset IS the
// LDS address: reads go through integer->address_space(3) casts so that no base add sits on the walk's
// dependent chain (witness_kernel traps if the assumption ever breaks).
typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
typedef uint32_t v2u32 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const v4u32 lds_cv4u32;
__device__ __forceinline__ uint32_t lds_u32(uint32_t off) { return *(lds_cu32 *)(uintptr_t)off; }
__device__ __forceinline__ uint4 lds_u128(uint32_t off) {
    const v4u32 v = *(lds_cv4u32 *)(uintptr_t)off;
    return make_uint4(v.x, v.y, v.z, v.w);
}

// delta lookup.  GTAB: the fused table did not fit the LDS budget and is read from global memory (it stays L2/MALL
// resident: every wave hammers the same few hundred KiB); same entry format, same byte offsets, ~10x the latency.
template <bool GTAB>
__device__ __forceinline__ uint32_t table_at(const WitnessArgs &a, uint32_t off) {
    if (GTAB) return a.table_image[off >> 2];
    return lds_u32(off);
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, s, 64));
    return v;
}

template <int D>
struct LaneRegs {
    uint32_t e[D];   // current fused entry of def d: bits 10.. = absolute table row of the CURRENT state
    uint32_t mx[D];  // running max of entries (reaching the dead row = an undefined transition)
    uint32_t sid_prev;
    uint32_t ov_row;  // D > 1: lowest row where two defs raise the same flag
};

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
    // can walk while that wave sits in its store burst behind HBM back-pressure.
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
                            if (it0 + i < nit) *reinterpret_cast<uint4 *>(gp) = v[i];
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
                            *reinterpret_cast<uint4 *>(mp) = make_uint4(0, 0, 0, 0);
                            mp += mstep;
                        }
                    } else {
                        for (uint32_t it = 0; it < GS / 8u; ++it) {
                            *reinterpret_cast<uint4 *>(mp) = masked_chunk<D>(it * 8u + mj0, mw, rec_base, chr_base, mb_base);
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
// Measured on MI355X (profiles/): with one wave doing everything, each tile's store burst
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
constexpr uint32_t kNoFix = 0xffffffffu;

__device__ __forceinline__ uint32_t lds_vol_u32(uint32_t off) { return *(volatile lds_cu32 *)(uintptr_t)off; }
__device__ __forceinline__ void lds_store_u32(uint32_t off, uint32_t v) {
    *(volatile __attribute__((address_space(3))) uint32_t *)(uintptr_t)off = v;
}
// wait until the tile counter at LDS offset `off` reaches `want`
__device__ __forceinline__ void ring_wait(uint32_t off, uint32_t want) {
    while ((int32_t)(lds_vol_u32(off) - want) < 0) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void ring_post(uint32_t off, uint32_t v) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // s_waitcnt lgkmcnt(0): the slot's LDS traffic is done
    lds_store_u32(off, v);
}

template <int D, int T>
__global__ __launch_bounds__(512) void witness_split_kernel(const WitnessArgs a, const uint32_t nslots) {
    using G = SplitGeom<D, T>;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t pairs = blockDim.x >> 7;  // walker waves 0..pairs-1, storer waves pairs..2*pairs-1
    const bool is_walker = wave < pairs;
    const uint32_t pair = is_walker ? wave : wave - pairs;
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();

    const uint32_t pair_bytes = nslots * G::kSlotBytes + G::kPairFixed;
    const uint32_t ring_base = a.table_bytes + pair * pair_bytes;
    const uint32_t prod_off = ring_base + nslots * G::kSlotBytes, cons_off = prod_off + 4u;
    const uint32_t scratch_off = a.table_bytes + pairs * pair_bytes + pair * 256u;  // 256 B per storer: LDS-DMA sink
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.table_image);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = threadIdx.x; i < a.table_bytes / 16u; i += blockDim.x) dst[i] = src[i];
        if (is_walker && lane == 0) { lds_store_u32(prod_off, 0); lds_store_u32(cons_off, 0); }
    }
    __syncthreads();

    const uint32_t M = a.M;
    const uint32_t ntiles = (M + T - 1u) / T;
    const uint32_t l7 = lane & 7u;
    uint32_t seq = 0;  // tiles handed over by this pair so far

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
                if (seq >= nslots) ring_wait(cons_off, seq - nslots + 1u);  // the storer has drained this slot
                if (stamp && lane == 0) stamp[1] = __builtin_amdgcn_s_memtime();

                // ---------------- walk + tag: lib.rs:804-888 ----------------
                TileBits tb = {0, 0, 0};
                const bool full = (t0 + T < min_n);
                if (a.debug & kDbgSplitNoWalk) {
                    // profiling only: no walk, the storer moves whatever the slot holds
                } else if (full)
                    tb = walk_tile<D, true, T>(L, act, a, chunks, 0, 0, t0);
                else
                    tb = walk_tile<D, false, T>(L, act, a, chunks, (int)n - (int)t0, (int)M - 1 - (int)t0, t0);

                auto rec_addr = [&](uint32_t i) { return slot + lane * 128u + (((i >> 2) ^ l7) << 4) + (i & 3u) * 4u; };
                // ---------------- undefined transition (lib.rs:817): rare slow path ----------------
                uint32_t newly = 0;
#pragma unroll
                for (int d = 0; d < D; ++d)
                    if (!((dead >> d) & 1u) && L.mx[d] >= a.dc[d].dead_entry) newly |= 1u << d;
                if (__any(newly != 0)) {
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        if ((newly >> d) & 1u) {
                            const uint32_t dead_state = a.dc[d].n_rows - 1u;
                            for (uint32_t p = 0; p < (uint32_t)T; ++p) {
                                const uint32_t s_p = lds_u32(rec_addr(p * D + d)) & 0xffffu;
                                const uint32_t s_n = (p + 1u < (uint32_t)T) ? (lds_u32(rec_addr((p + 1u) * D + d)) & 0xffffu)
                                                                            : ((L.e[d] >> kNextShift) - a.dc[d].row_base);
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
                            accept |= (((L.e[d] >> kNextShift) - a.dc[d].row_base) == a.dc[d].accepted_state ? 1u : 0u) << d;
                    }
                }
                // ---------------- reveal masks: lib.rs:598-764 ----------------
                TileMasks tm = tile_masks<T>(tb, mc, t0, tile_is_exact(t0, n, M, T), rows_below(t0, n));
                if (!active) { tm.mask = 0; tm.fix = 0; }
                *(__attribute__((address_space(3))) v2u32 *)(uintptr_t)(slot + G::kSlotHdr + lane * 8u) =
                    v2u32{(uint32_t)tm.mask, tm.fix ? tm.fix_start : kNoFix};
                if (__any(tm.mask != 0)) {
#pragma unroll
                    for (int i = 0; i < CPT; ++i)
                        *(__attribute__((address_space(3))) v4u32 *)(uintptr_t)(slot + G::kSlotChr + lane * T + 16u * i) =
                            v4u32{act[i].x, act[i].y, act[i].z, act[i].w};
                }
                ring_post(prod_off, seq + 1u);
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
            constexpr uint32_t kBlk = 64u;
            constexpr uint32_t kTouchRows = 128u, kTouchAhead = 5u * 128u;
            const uint32_t n_s = active ? min(a.lens[b], M) : M;
            const uint32_t last_line = n_s ? ((n_s - 1u) & ~127u) : 0u;
            const uint8_t *tptr = a.chars + (size_t)(active ? b : a.B - 1u) * a.stride;
            auto touch = [&](uint32_t row) {
                if (!(a.debug & kDbgNoTouch)) {
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
                uint64_t fixm = __ballot(hdr.y != kNoFix);
                if (fixm) {  // an earlier optimistic end_mask = 1 turned out wrong: zero those masked rows (rare)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    while (fixm) {
                        const int j = __ffsll((unsigned long long)fixm) - 1;
                        fixm &= fixm - 1;
                        const uint32_t fs = (uint32_t)__builtin_amdgcn_readlane((int)hdr.y, j);
                        // rows already in memory: everything below this 64-row block
                        uint16_t *mrow = a.masked + (size_t)(b0 + j) * a.msk_pitch;
                        for (uint32_t r = (fs & ~63u) + lane; r < blk0; r += 64u)
                            if (r >= fs) mrow[r] = 0;
                        // rows of this block still held in mk (earlier tiles of the block)
                        if (t0 > blk0 && fs < t0) {
                            const uint32_t r0 = blk0 + mw * 8u;  // first row of this lane's chunk
#pragma unroll
                            for (int it = 0; it < 8; ++it) {
                                if ((uint32_t)it == ((uint32_t)j >> 3) && mj0 == ((uint32_t)j & 7u) && r0 < t0 && r0 + 8u > fs) {
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
                    if (!(a.debug & kDbgSkipRecords)) {
#pragma unroll
                        for (uint32_t it = 0; it < 8u; ++it) {
                            if (whole || (b0 + it * 8u + js0 < a.B && w < lim)) *reinterpret_cast<uint4 *>(gp) = v[it];
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
                ring_post(cons_off, seq + 1u);  // every LDS read of the slot has returned; the stores may still be in flight
                if (t0 % kTouchRows == 0) touch(t0 + kTouchAhead + kTouchRows);
                if (blk_last && !(a.debug & kDbgSkipMasked)) {
                    unsigned char *mp = reinterpret_cast<unsigned char *>(a.masked + (size_t)(b0 + mj0) * a.msk_pitch + blk0 + mw * 8u);
                    const size_t mstep = (size_t)8u * a.msk_pitch * 2u;
                    const bool blk_whole = (b0 + 64u <= a.B) && (blk0 + kBlk <= M);
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        if (blk_whole || (b0 + (uint32_t)it * 8u + mj0 < a.B && blk0 + mw * 8u < M)) *reinterpret_cast<uint4 *>(mp) = mk[it];
                        mp += mstep;
                    }
                }
                if (stamp && lane == 0) stamp[6] = __builtin_amdgcn_s_memtime();
            }
        }
    }
}

// =============================================================================================
// Position-major kernel (layout 1): records [ceil(M/4)][D][B][4] u32, masked [ceil(M/8)][B][8] u16.
//
// With one lane per string, four consecutive rows of a lane are 16*D contiguous bytes and the 64 lanes of a wave are
// 64 consecutive strings: every store is a full, contiguous 1-KiB (D=1) run written straight from the walker's
// registers — no LDS transpose, no mover wave — and at any moment the whole chip writes into one compact slab of the
// output (rows 4q..4q+3 of all strings = 1 MiB at B = 65536).  A compact write window is what the HBM write path
// rewards: 6.5 TB/s vs 4.3-5.2 TB/s for the string-major comb (tools/fillprobe, tools/wpattern2).
//
// The walker's in-order vmcnt would make any wait for an input load also wait for every store issued before it, so
// the walker issues no loads at all: a LOADER wave per walker streams the strings' bytes into an LDS ring with LDS-DMA
// (global_load_lds_dwordx4: no VGPRs, kRing tiles in flight, counted s_waitcnt) and the walker picks its 64 bytes per
// tile up with four ds_read_b128.
// =============================================================================================
constexpr uint32_t kPmTileBytes = 64u * 64u;  // 64 strings x 64 input bytes per tile

__device__ __forceinline__ void store16(unsigned char *p, const uint4 &v, const bool nt) {
    if (nt) __builtin_nontemporal_store(v4u32{v.x, v.y, v.z, v.w}, reinterpret_cast<v4u32 *>(p));
    else *reinterpret_cast<uint4 *>(p) = v;
}

// Where a walker's finished rows go: straight to memory from its registers.  quad(d, p, ..) stores four rows of def d
// (16 B per lane, 1 KiB contiguous per wave) into its plane of [ceil(M/4)][D][B][4]; row(p) lets the previous tile's masked
// rows leave one 16-byte piece every 8 rows.  (The walk functions take the sink as a policy: a variant that handed the
// rows to a third "storer" wave through an LDS out-ring, so that the walker issued no vector-memory instruction at all, was
// built and measured in round 1 — every global store does cost the issuing wave 75-125 cycles, but the ds_write_b128 +
// hand-over cost the walker as much, and where all walker slots are busy the launch is bound by the memory system's mixed
// read/write rate anyway: 97 vs 89 us on the headline workload, 3.41 vs 3.39 ms on cfg 4.  Dropped)
template <int D>
struct GlobalSink {
    static constexpr bool kSidq = true;
    unsigned char *rp;
    size_t plane, rstep;
    bool do_store, nt_rec, nt_msk;
    const uint4 (&pend)[8];
    unsigned char *pend_mp;
    size_t mstep;
    bool pend_store;
    __device__ __forceinline__ void quad(const int d, const int p, const bool full, const int mrem, const uint4 &v) {
        // quads that start at or beyond row M do not exist in [ceil(M/4)][D][B][4]
        if (do_store && (full || (p & ~3) <= mrem)) store16(rp + (size_t)d * plane, v, nt_rec);
        if (d == D - 1) rp += rstep;
    }
    __device__ __forceinline__ void row(const int p) {
        // the PREVIOUS tile's masked rows leave one 16-byte piece every 8 rows instead of as a burst of 8 stores at the
        // tile boundary (the burst filled the store queue and stalled the in-order walk: 98.7 -> 93.8 us)
        if (D == 1 && (p & 7) == 5 && pend_store) store16(pend_mp + (size_t)(p >> 3) * mstep, pend[p >> 3], nt_msk);
    }
};

typedef __attribute__((address_space(3))) const uint16_t lds_cu16;
__device__ __forceinline__ uint32_t lds_u16(uint32_t off) { return *(lds_cu16 *)(uintptr_t)off; }
// HALF table (hrx_lane.h): address of entry (row of `e`, byte c) from e and c2 = c << 1 — one v_perm_b32:
// byte 0 = c2.byte0 = (c & 127) << 1, byte 1 = e.byte0 = row, byte 2 = c2.byte1 = c >> 7, byte 3 = 0
__device__ __forceinline__ uint32_t half_next_addr(uint32_t e, uint32_t c2) { return __builtin_amdgcn_perm(e, c2, 0x0c010400u); }
__device__ __forceinline__ uint32_t half_tag(uint32_t e) { return ((e >> 8) & 0x3fu) | ((e >> 14) << 8); }  // -> the narrow format's 10-bit tag

template <int D, bool FULL, bool GTAB, bool HALF, class Sink>
__device__ __forceinline__ TileBits walk_tile_pm(LaneRegs<D> &L, const uint4 (&cq)[4], const WitnessArgs &a, Sink &sink, int rem, int mrem,
                                                 uint32_t t0, uint32_t (&sidq)[16], uint32_t (&acc_state)[D]) {
    uint32_t st[2] = {0, 0}, en1[2] = {0, 0}, ch[2] = {0, 0};
    uint32_t rbuf[D][4];
    const uint32_t cw[16] = {cq[0].x, cq[0].y, cq[0].z, cq[0].w, cq[1].x, cq[1].y, cq[1].z, cq[1].w,
                             cq[2].x, cq[2].y, cq[2].z, cq[2].w, cq[3].x, cq[3].y, cq[3].z, cq[3].w};
    uint32_t e1[D], e2[D], raw[D];
#pragma unroll
    for (int d = 0; d < D; ++d) e1[d] = e2[d] = L.e[d];

    auto post = [&](const int p, const uint32_t (&es)[D], const uint32_t (&et)[D]) {
        uint32_t sid = 0, stn = 0, enn = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            uint32_t state = HALF ? (es[d] & 0xffu) - (d ? a.dc[d].half_row_base : 0u) : (es[d] >> kNextShift) - (d ? a.dc[d].row_base : 0u);
            uint32_t tag = HALF ? half_tag(et[d]) : et[d] & kTagMask;
            if (!FULL) {
                if (HALF && p > rem) state = a.dc[d].dummy_state;  // the HALF image has no dummy row (lib.rs:413)
                if (p >= mrem) tag &= ~kTagEnd;
                if (p == rem) acc_state[d] = state;  // the state at row n (lib.rs:437-457)
            }
            rbuf[d][p & 3] = state | (tag << 16);
            // four rows of def d of this string: 16 bytes, a 1-KiB contiguous run across the wave
            if ((p & 3) == 3) sink.quad(d, p, FULL, mrem, make_uint4(rbuf[d][0], rbuf[d][1], rbuf[d][2], rbuf[d][3]));
            if (!FULL || HALF) L.mx[d] = max(L.mx[d], et[d]);  // HALF: an undefined transition is a marked entry, not an absorbing row
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
        if (Sink::kSidq) sidq[p >> 2] |= sid << (8 * (p & 3));  // the tile's substr-id sums, one byte per row (masked rows need them)
        sink.row(p);
    };
#pragma unroll
    for (int i = 0; i < 16; ++i) sidq[i] = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = q * 4 + k;
            const uint32_t c4 = ((cw[q] >> (8 * k)) & 0xffu) << (HALF ? 1 : 2);
#pragma unroll
            for (int d = 0; d < D; ++d)  // delta(state, byte): lib.rs:810
                raw[d] = HALF ? lds_u16(half_next_addr(e1[d], c4)) : table_at<GTAB>(a, (e1[d] & ~kTagMask) | c4);
            if (p > 0) {
                post(p - 1, e2, e1);
                asm volatile("" : "+v"(st[(p - 1) >> 5]), "+v"(en1[(p - 1) >> 5]), "+v"(ch[(p - 1) >> 5]), "+v"(L.sid_prev));
                if (Sink::kSidq) asm volatile("" : "+v"(sidq[(p - 1) >> 2]));
                if (!FULL || HALF) {
#pragma unroll
                    for (int d = 0; d < D; ++d) asm volatile("" : "+v"(L.mx[d]));
                }
                if (!FULL) {
#pragma unroll
                    for (int d = 0; d < D; ++d) asm volatile("" : "+v"(acc_state[d]));
                }
                if (D > 1) asm volatile("" : "+v"(L.ov_row));  // or the 64 per-row flag counts stay live until the tile end
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = 0; d < D; ++d) {
                e2[d] = e1[d];
                // rows >= n: lib.rs:404-418 (HALF: any valid row with an empty tag; post() writes the dummy state)
                e1[d] = (FULL || p < rem) ? raw[d] : (HALF ? a.dc[d].half_row_base : a.dc[d].dummy_entry);
            }
        }
    }
    post(63, e2, e1);
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

// ---------------------------------------------------------------------------------------------
// WIDE-table walk (hrx_lane.h): one ds_read_b64 per row and def returns the chain word AND the finished record, so a
// row costs, beyond the lookups,  v_add3 (per-row sums of substr ids and flag counts over the defs, straight from the
// chain words) + v_bfe (substr id) + 2 shifts + 2 v_alignbit (start / end bit into the tile bitvectors) +
// v_cmp/v_addc (id-changed bit) + 1 v_lshl_or (the id byte kept for the masked rows)  —  ~14 VALU at D = 3 against
// ~65 for the narrow entry format, which made the D = 3 walk issue-bound (a wave64 VALU op occupies its SIMD for
// 4 cycles).  Two defs flagging the same row only set tile_ov != 0 here; the exact row is found by the tile re-walk.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) const v2u32 lds_cv2u32;
__device__ __forceinline__ uint2 lds_u64(uint32_t off) {
    const v2u32 v = *(lds_cv2u32 *)(uintptr_t)off;
    return make_uint2(v.x, v.y);
}

template <int D, bool FULL, class Sink>
__device__ __forceinline__ TileBits walk_tile_pm_wide(LaneRegs<D> &L, const uint4 (&cq)[4], const WitnessArgs &a, Sink &sink, int rem, int mrem,
                                                      uint32_t &tile_ov, uint32_t (&sidq)[16], uint32_t (&acc_state)[D]) {
    uint32_t st[2] = {0, 0}, en1[2] = {0, 0}, ch[2] = {0, 0};
    uint32_t rbuf[D][4];
    uint32_t ov = 0;
    // bytes >= 128 have no column: they are masked here and the tile is re-walked by the caller
    const uint32_t cw[16] = {cq[0].x & 0x7f7f7f7fu, cq[0].y & 0x7f7f7f7fu, cq[0].z & 0x7f7f7f7fu, cq[0].w & 0x7f7f7f7fu,
                             cq[1].x & 0x7f7f7f7fu, cq[1].y & 0x7f7f7f7fu, cq[1].z & 0x7f7f7f7fu, cq[1].w & 0x7f7f7f7fu,
                             cq[2].x & 0x7f7f7f7fu, cq[2].y & 0x7f7f7f7fu, cq[2].z & 0x7f7f7f7fu, cq[2].w & 0x7f7f7f7fu,
                             cq[3].x & 0x7f7f7f7fu, cq[3].y & 0x7f7f7f7fu, cq[3].z & 0x7f7f7f7fu, cq[3].w & 0x7f7f7f7fu};
    uint32_t lo[D], plo[D], phi[D];   // lo: chain word after the newest row; plo/phi: chain word and record of the row being posted
#pragma unroll
    for (int d = 0; d < D; ++d) { lo[d] = plo[d] = L.e[d]; phi[d] = 0; }

    auto post = [&](const int p) {    // row p: chain words plo[], records phi[]
        uint32_t T;
        if (D == 1) T = plo[0];
        else if (D == 2) T = plo[0] + plo[1];
        else T = plo[0] + plo[1] + plo[D - 1];
        if (!FULL) {
            if (p >= mrem) T &= ~(3u << kWideEndShift);   // end_enable of row M-1 is never assigned (lib.rs:501)
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            uint32_t rec = phi[d];
            if (!FULL) {
                if (p >= mrem) rec &= ~(1u << 25);
            }
            rbuf[d][p & 3] = rec;
            // four rows of def d of this string: 16 bytes, a 1-KiB contiguous run across the wave
            if ((p & 3) == 3) sink.quad(d, p, FULL, mrem, make_uint4(rbuf[d][0], rbuf[d][1], rbuf[d][2], rbuf[d][3]));
        }
        const uint32_t sid = (T >> kWideSidShift) & 0xffu;
        const uint32_t F = T >> kWideStartShift;          // bits 0..1 start count, 2..3 end count
        if (D > 1) ov |= F & 0xau;                        // a count of 2 or 3: two defs flag the same row
        st[p >> 5] = __builtin_amdgcn_alignbit(F, st[p >> 5], 1);
        en1[p >> 5] = __builtin_amdgcn_alignbit(T >> kWideEndShift, en1[p >> 5], 1);
        // ch = (ch << 1) | (sid != sid_prev): bits arrive in reverse row order, undone once per word below
        asm volatile("v_cmp_ne_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(ch[p >> 5]) : "v"(sid), "v"(L.sid_prev) : "vcc");
        L.sid_prev = sid;
        if (Sink::kSidq) sidq[p >> 2] |= sid << (8 * (p & 3));
        sink.row(p);
    };
#pragma unroll
    for (int i = 0; i < 16; ++i) sidq[i] = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = q * 4 + k;
            const uint32_t c8 = ((cw[q] >> (8 * k)) & 0xffu) << 3;
            uint2 raw[D];
#pragma unroll
            for (int d = 0; d < D; ++d) raw[d] = lds_u64((lo[d] & kWideRowMask) | c8);   // delta(state, byte): lib.rs:810
            if (p > 0) {
                post(p - 1);
                asm volatile("" : "+v"(st[(p - 1) >> 5]), "+v"(en1[(p - 1) >> 5]), "+v"(L.sid_prev));
                if (Sink::kSidq) asm volatile("" : "+v"(sidq[(p - 1) >> 2]));
                if (D > 1) asm volatile("" : "+v"(ov));
                if (!FULL) {   // or the selects of all 64 rows are deferred to the tile end with every lookup result kept live (300 spills at D = 3)
#pragma unroll
                    for (int d = 0; d < D; ++d) asm volatile("" : "+v"(L.mx[d]), "+v"(acc_state[d]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const uint32_t prev = lo[d];             // chain word after row p-1: its row field is the state at row p
                if (FULL) {
                    lo[d] = raw[d].x;
                    phi[d] = raw[d].y;
                } else {
                    const bool live = p < rem;
                    const uint32_t state_here = ((prev >> kWideRowShift) & 0xffu) - a.dc[d].row_base;
                    if (p == rem) acc_state[d] = state_here;                       // the state at row n (lib.rs:437-457)
                    lo[d] = live ? raw[d].x : a.dc[d].dummy_entry;                 // rows >= n: lib.rs:404-418
                    phi[d] = live ? raw[d].y : (p == rem ? state_here : (a.dc[d].dummy_entry >> kWideRowShift) - a.dc[d].row_base);
                    L.mx[d] = live ? raw[d].x : L.mx[d];                           // last real chain word (dead-row check)
                }
                plo[d] = lo[d];
            }
        }
    }
    post(63);
#pragma unroll
    for (int d = 0; d < D; ++d) {
        L.e[d] = lo[d];
        if (FULL) L.mx[d] = lo[d];
    }
    tile_ov = ov;
    TileBits tb;
    tb.st = (uint64_t)st[0] | ((uint64_t)st[1] << 32);
    tb.en1 = (uint64_t)en1[0] | ((uint64_t)en1[1] << 32);
    tb.ch = (uint64_t)__builtin_bitreverse32(ch[0]) | ((uint64_t)__builtin_bitreverse32(ch[1]) << 32);
    return tb;
}

template <int D, bool GTAB, bool WIDE, bool HALF = false>
__global__ __launch_bounds__(512) void witness_pm_kernel(const WitnessArgs a, const uint32_t nring) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t pairs = blockDim.x >> 7;  // walker waves 0..pairs-1, loader waves pairs..2*pairs-1
    const bool is_walker = wave < pairs;
    const uint32_t pair = is_walker ? wave : wave - pairs;
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();

    // ring + the walker's 4-KiB scratch (HALF: none, its slow path re-walks out of registers) + counters
    const uint32_t pair_bytes = nring * kPmTileBytes + (HALF ? 0u : kPmTileBytes) + 16u;
    const uint32_t tab_bytes = GTAB ? 0u : HALF ? a.half_bytes : a.table_bytes;
    const uint32_t ring_base = tab_bytes + pair * pair_bytes;
    const uint32_t scratch_off = ring_base + nring * kPmTileBytes;
    const uint32_t ready_off = scratch_off + (HALF ? 0u : kPmTileBytes), freed_off = ready_off + 4u;
    const uint32_t M = a.M, B = a.B;
    const uint32_t ntiles = (M + 63u) >> 6;
    uint32_t seq = 0;
    const uint32_t g_first = blockIdx.x * pairs + pair, g_stride = gridDim.x * pairs;
    // The loaders request their pair's first input tile BEFORE the table is staged: the HBM round trip (~2 us) then runs
    // under the staging instead of after it.
    uint32_t first_len = M;   // ... and the walkers their first group's lengths
    if (is_walker && g_first < a.n_groups && g_first * 64u + lane < B) first_len = a.lens[g_first * 64u + lane];
    uint4 first_tile[4];
    if (!is_walker && g_first < a.n_groups) {
        const bool in_pm0 = (a.layout & 2u) != 0;
        const uint32_t bl = min(g_first * 64u + lane, B - 1u);
        const uint8_t *cptr = in_pm0 ? a.chars + (size_t)bl * 16u : a.chars + (size_t)bl * a.stride;
        const uint32_t row_cap0 = (uint32_t)a.stride - 16u;
        const size_t cmul0 = (a.debug & kDbgInputFromL2) ? (size_t)0 : in_pm0 ? (size_t)B : (size_t)1;
#pragma unroll
        for (uint32_t i = 0; i < 4u; ++i) first_tile[i] = *reinterpret_cast<const uint4 *>(cptr + (size_t)min(16u * i, row_cap0) * cmul0);
    }
    {
        const uint4 *src = WIDE ? reinterpret_cast<const uint4 *>(a.wide_image)
                                : HALF ? reinterpret_cast<const uint4 *>(a.half_image) : reinterpret_cast<const uint4 *>(a.table_image);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        if (!GTAB)
            for (uint32_t i = threadIdx.x; i < tab_bytes / 16u; i += blockDim.x) dst[i] = src[i];
        if (is_walker && lane == 0) { lds_store_u32(ready_off, 0); lds_store_u32(freed_off, 0); }
    }
    __syncthreads();

    if (!is_walker) {
        // ================================ loader ================================
        // string-major input: string b at chars + b*stride; position-major input: 16-byte chunk i of string b at
        // chars + (i*B + b)*16, so one load instruction reads 1 KiB contiguous (coalesced, compact read window).
        //
        // The loader runs RT tiles (RT*4 KiB of its pair's input, 16 B per lane per load) ahead of the walker, in its own
        // registers (192 VGPRs at D = 1 that the kernel owns anyway), over the flattened (group, tile) sequence of the
        // pair: at M <= 1024 practically the whole input of a group is requested in one burst at the start, and the next
        // group's bytes are on their way long before the walker gets there.  It issues nothing but these loads, so the
        // counted s_waitcnt vmcnt(4*(RT-1)) for the oldest tile is exact.  Bytes at or beyond a string's length are
        // read (inside the string's own stride) but never trusted.
        constexpr uint32_t RT = D == 1 ? 12u : 8u;
        const bool in_pm = (a.layout & 2u) != 0;
        const uint32_t my_groups = g_first < a.n_groups ? (a.n_groups - g_first + g_stride - 1u) / g_stride : 0u;
        const uint32_t total = my_groups * ntiles;
        const uint32_t row_cap = (uint32_t)a.stride - 16u;  // last 16-byte chunk that exists for every string
        const size_t cmul = in_pm ? (size_t)B : (size_t)1;  // byte offset of chunk-start row r: r * cmul
        const size_t cmul_eff = (a.debug & kDbgInputFromL2) ? (size_t)0 : cmul;  // (4: profiling only, every tile re-reads the hot first lines)
        uint4 buf[RT * 4u];
        auto issue = [&](const uint32_t q, const uint32_t k) {  // tile q of the pair's sequence -> register tile k
            const uint32_t g = g_first + (q / ntiles) * g_stride, t = q % ntiles;
            const uint32_t bl = min(g * 64u + lane, B - 1u);
            const uint8_t *cptr = in_pm ? a.chars + (size_t)bl * 16u : a.chars + (size_t)bl * a.stride;
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) {
                const size_t off = (size_t)min(t * 64u + 16u * i, row_cap) * cmul_eff;
                buf[k * 4u + i] = *reinterpret_cast<const uint4 *>(cptr + off);
            }
        };
        // The pair's FIRST tile travels alone: requested together with the rest, it queues behind the whole chip's opening
        // burst (~48 MiB) and reaches the walker ~10 us into the launch (in-kernel stamps, tools/kbench) instead of ~1.5.
        if (total > 0) {   // (requested before the table staging; see the kernel's prologue)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) {
                uint4 v = first_tile[i];
                asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
                *(__attribute__((address_space(3))) v4u32 *)(uintptr_t)(ring_base + i * 1024u + lane * 16u) = v4u32{v.x, v.y, v.z, v.w};
            }
            ring_post(ready_off, 1u);
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
                    if (sq >= nring) ring_wait(freed_off, sq - nring + 1u);  // the walker has read this slot
                    const uint32_t slot = ring_base + (sq % nring) * kPmTileBytes;
                    // tile sq was requested RT tiles ago; RT-1 younger tiles (4 loads each) may still be in flight
                    if (sq + RT <= total) {
                        if (RT == 12u) asm volatile("s_waitcnt vmcnt(44)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tail of the sequence: nothing younger is being issued
                    }
#pragma unroll
                    for (uint32_t i = 0; i < 4u; ++i) {
                        uint4 v = buf[k * 4u + i];
                        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));  // after the counted wait, not before
                        *(__attribute__((address_space(3))) v4u32 *)(uintptr_t)(slot + i * 1024u + lane * 16u) = v4u32{v.x, v.y, v.z, v.w};
                    }
                    ring_post(ready_off, sq + 1u);
                    if (sq + RT < total) issue(sq + RT, k);
                }
            }
        }
        return;
    }

    for (uint32_t g = g_first; g < a.n_groups; g += g_stride) {
        const uint32_t b0 = g * 64u;
        const uint32_t b = b0 + lane;
        const bool active = b < B;
        const uint32_t n_raw = g == g_first ? first_len : (active ? a.lens[b] : M);
        const bool badlen = n_raw > M;
        const uint32_t n = badlen ? M : n_raw;

        {
            // ================================ walker ================================
            const uint32_t min_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_min_u32(n));
            LaneRegs<D> L;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                L.e[d] = HALF ? a.dc[d].half_row_base + a.dc[d].first_state : a.dc[d].first_entry;  // states[d][0] = first_state_val: lib.rs:807
                L.mx[d] = 0;
            }
            L.sid_prev = 0;
            L.ov_row = 0xffffffffu;
            MaskCarry mc = {0, 0, 0, 0};
            uint32_t dead = 0, accept = 0;
            uint32_t err_pos[D], err_state[D], err_char[D], acc_state[D];
#pragma unroll
            for (int d = 0; d < D; ++d) {
                err_pos[d] = err_state[d] = err_char[d] = 0;
                acc_state[d] = a.dc[d].first_state;  // n == 0
            }
            const uint32_t bc = active ? b : B - 1u;  // idle lanes shadow the last string (their stores are masked off)
            unsigned char *rp = reinterpret_cast<unsigned char *>(a.records) + (size_t)bc * 16u * ((a.debug & kDbgInterleavedDefs) ? D : 1);
            // (kDbgFixedLines, profiling only: every quad / octet of a string lands on the first one — same store instructions, no new lines or pages)
            const size_t rstep = (a.debug & kDbgFixedLines) ? (size_t)0 : (size_t)B * 16u * D;  // one quad of rows further: [M/4][D][B][4]
            unsigned char *mp = reinterpret_cast<unsigned char *>(a.masked) + (size_t)bc * 16u;
            const size_t mstep = (a.debug & kDbgFixedLines) ? (size_t)0 : (size_t)B * 16u;      // 8 rows further: [M/8][B][8]
            uint4 pend[8];                             // the previous tile's masked rows, not yet stored
            unsigned char *pend_mp = mp;
            bool have_pend = false;
#pragma unroll
            for (int k = 0; k < 8; ++k) pend[k] = make_uint4(0, 0, 0, 0);

            for (uint32_t t = 0; t < ntiles; ++t, ++seq) {
                const uint32_t t0 = t << 6;
                const uint32_t slot = ring_base + (seq % nring) * kPmTileBytes;
                ring_wait(ready_off, seq + 1u);
                uint4 cq[4];
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i) cq[i] = lds_u128(slot + i * 1024u + lane * 16u);
                ring_post(freed_off, seq + 1u);

                uint32_t e_start[D];
#pragma unroll
                for (int d = 0; d < D; ++d) e_start[d] = L.e[d];
                uint32_t sidq[16];
                TileBits tb;
                const bool full = (t0 + 64u < min_n);
                const bool do_store = active && !(a.debug & kDbgSkipRecords);
                const bool pend_store = active && have_pend && !(a.debug & kDbgSkipMasked);
                uint32_t tile_ov = 0, hb = 0;   // WIDE: flag-overlap seen in the tile; bytes >= 128 among the tile's live rows
                // [ceil(M/4)][D][B][4]: one def's quads of all strings (kDbgInterleavedDefs, profiling: [M/4][B][D][4])
                GlobalSink<D> sink{rp, (a.debug & kDbgInterleavedDefs) ? (size_t)16u : (size_t)B * 16u, rstep, do_store, (a.debug & kDbgNtRecords) != 0, (a.debug & kDbgNtMasked) != 0,
                                   pend, pend_mp, mstep, pend_store};
                if (WIDE) {
                    const uint32_t cwl[16] = {cq[0].x, cq[0].y, cq[0].z, cq[0].w, cq[1].x, cq[1].y, cq[1].z, cq[1].w,
                                              cq[2].x, cq[2].y, cq[2].z, cq[2].w, cq[3].x, cq[3].y, cq[3].z, cq[3].w};
                    if (full) {
                        tb = walk_tile_pm_wide<D, true>(L, cq, a, sink, 0, 0, tile_ov, sidq, acc_state);
#pragma unroll
                        for (int q = 0; q < 16; ++q) hb |= cwl[q];
                        hb &= 0x80808080u;
                    } else {
                        tb = walk_tile_pm_wide<D, false>(L, cq, a, sink, (int)n - (int)t0, (int)M - 1 - (int)t0, tile_ov, sidq, acc_state);
                        const uint32_t live_rows = n > t0 ? min(n - t0, 64u) : 0u;
#pragma unroll
                        for (int q = 0; q < 16; ++q) {   // bytes at or beyond the string's length are not trusted
                            const uint32_t nb = live_rows > 4u * q ? min(live_rows - 4u * q, 4u) : 0u;
                            hb |= cwl[q] & (nb >= 4u ? 0xffffffffu : ((1u << (8u * nb)) - 1u));
                        }
                        hb &= 0x80808080u;
                    }
                } else if (full)
                    tb = walk_tile_pm<D, true, GTAB, HALF>(L, cq, a, sink, 0, 0, t0, sidq, acc_state);
                else
                    tb = walk_tile_pm<D, false, GTAB, HALF>(L, cq, a, sink, (int)n - (int)t0, (int)M - 1 - (int)t0, t0, sidq, acc_state);
                rp = sink.rp;

                // ---------------- undefined transition (lib.rs:817): rare slow path, re-walk the tile ----------------
                uint32_t newly = 0;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    // WIDE: the dead row absorbs, so the last real chain word tells; a byte >= 128 has no column and was
                    // walked through its masked alias, so such a tile is re-walked as well
                    const bool hit = WIDE ? ((L.mx[d] & kWideRowMask) == a.dc[d].dead_entry || hb != 0)
                                          : HALF ? L.mx[d] >= kHalfDead : L.mx[d] >= a.dc[d].dead_entry;
                    if (!((dead >> d) & 1u) && hit) newly |= 1u << d;
                }
                if (HALF && __any(newly != 0)) {
                    // no scratch area in this variant (a 256-state table leaves 32 KiB of LDS for all the rings): the tile is
                    // re-walked out of the byte registers, fully unrolled
                    const uint32_t cwl[16] = {cq[0].x, cq[0].y, cq[0].z, cq[0].w, cq[1].x, cq[1].y, cq[1].z, cq[1].w,
                                              cq[2].x, cq[2].y, cq[2].z, cq[2].w, cq[3].x, cq[3].y, cq[3].z, cq[3].w};
                    const uint32_t live_rows = n > t0 ? min(n - t0, 64u) : 0u;
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        if ((newly >> d) & 1u) {
                            uint32_t e = e_start[d];
                            bool found = false;
#pragma unroll
                            for (int p = 0; p < 64; ++p) {
                                const uint32_t c = (cwl[p >> 2] >> (8 * (p & 3))) & 0xffu;
                                const uint32_t nx = lds_u16(half_addr(e & 0xffu, c));
                                if (!found && (uint32_t)p < live_rows && nx >= kHalfDead) {
                                    err_pos[d] = t0 + (uint32_t)p;
                                    err_state[d] = (e & 0xffu) - a.dc[d].half_row_base;
                                    err_char[d] = c;
                                    found = true;
                                }
                                e = nx;
                            }
                            dead |= 1u << d;
                        }
                    }
                } else if (__any(newly != 0)) {
                    // the tile's bytes go to this walker's LDS scratch so that the re-walk can index them at run time
#pragma unroll
                    for (uint32_t i = 0; i < 4u; ++i)
                        *(__attribute__((address_space(3))) v4u32 *)(uintptr_t)(scratch_off + lane * 64u + i * 16u) =
                            v4u32{cq[i].x, cq[i].y, cq[i].z, cq[i].w};
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        if ((newly >> d) & 1u) {
                            uint32_t e = e_start[d];
                            const uint32_t live_rows = n > t0 ? min(n - t0, 64u) : 0u;
                            bool found = false;
                            for (uint32_t p = 0; p < live_rows; ++p) {
                                const uint32_t c = smem[scratch_off + lane * 64u + p];
                                uint32_t nx;
                                bool bad;
                                if (WIDE) {
                                    nx = c < 128u ? lds_u32((e & kWideRowMask) | (c << 3)) : a.dc[d].dead_entry;
                                    bad = (nx & kWideRowMask) == a.dc[d].dead_entry;
                                } else {
                                    nx = table_at<GTAB>(a, (e & ~kTagMask) | (c << 2));
                                    bad = nx >= a.dc[d].dead_entry;
                                }
                                if (bad) {
                                    err_pos[d] = t0 + p;
                                    err_state[d] = (WIDE ? ((e >> kWideRowShift) & 0xffu) : (e >> kNextShift)) - a.dc[d].row_base;
                                    err_char[d] = c;
                                    found = true;
                                    break;
                                }
                                e = nx;
                            }
                            if (found || !WIDE) dead |= 1u << d;
                        }
                    }
                }
                // ---------------- WIDE: two defs flagged the same row somewhere in this tile: find the row (rare) ----------------
                if (WIDE && D > 1 && __any(tile_ov != 0 && L.ov_row == 0xffffffffu)) {
#pragma unroll
                    for (uint32_t i = 0; i < 4u; ++i)
                        *(__attribute__((address_space(3))) v4u32 *)(uintptr_t)(scratch_off + lane * 64u + i * 16u) =
                            v4u32{cq[i].x, cq[i].y, cq[i].z, cq[i].w};
                    if (tile_ov != 0 && L.ov_row == 0xffffffffu) {
                        uint32_t e[D];
#pragma unroll
                        for (int d = 0; d < D; ++d) e[d] = e_start[d];
                        const uint32_t live_rows = n > t0 ? min(n - t0, 64u) : 0u;
                        for (uint32_t p = 0; p < live_rows && L.ov_row == 0xffffffffu; ++p) {
                            const uint32_t c = smem[scratch_off + lane * 64u + p] & 0x7fu;
                            uint32_t T = 0;
#pragma unroll
                            for (int d = 0; d < D; ++d) {
                                e[d] = lds_u32((e[d] & kWideRowMask) | (c << 3));
                                T += e[d];
                            }
                            if (t0 + p + 1u >= M) T &= ~(3u << kWideEndShift);
                            const uint32_t F = T >> kWideStartShift;
                            if (F & 2u) L.ov_row = t0 + p;                       // two is_start flags on row p
                            else if (F & 8u) L.ov_row = t0 + p + 1u;             // two is_end flags on row p+1
                        }
                    }
                }
                // ---------------- accept state: the state at row n (lib.rs:437-457) ----------------
                if (!full && n == t0 + 64u && t + 1 == ntiles) {  // n == M: row n does not exist, s[n] is the live state
#pragma unroll
                    for (int d = 0; d < D; ++d)
                        acc_state[d] = HALF ? (L.e[d] & 0xffu) - a.dc[d].half_row_base
                                            : (WIDE ? ((L.e[d] >> kWideRowShift) & 0xffu) : (L.e[d] >> kNextShift)) - a.dc[d].row_base;
                }
                // ---------------- reveal masks: lib.rs:598-764 ----------------
                TileMasks tm = tile_masks<64>(tb, mc, t0, tile_is_exact(t0, n, M), rows_below(t0, n));
                if (!active) { tm.mask = 0; tm.fix = 0; }
                // An earlier optimistic end_mask = 1 turned out wrong: zero those masked rows (rare with real definitions; a
                // random DFA like cfg 5's takes this path every few tiles, and there each 16-byte piece re-written in a line
                // that has left L2 is a read-modify-write at the memory: measured 521 vs 357 us with the fix-ups skipped;
                // a per-lane variant that zeroes whole octets with 16-byte stores was no better — 558 us).
                uint64_t fixm = __ballot(tm.fix != 0);
                if (a.debug & kDbgSkipFixups) fixm = 0;  // profiling only: skip the fix-ups
                while (fixm) {
                    const int j = __ffsll((unsigned long long)fixm) - 1;
                    fixm &= fixm - 1;
                    const uint32_t fs = (uint32_t)__builtin_amdgcn_readlane((int)tm.fix_start, j);
                    const uint32_t bj = b0 + (uint32_t)j;
                    for (uint32_t r = fs + lane; r < t0; r += 64u)
                        a.masked[((size_t)(r >> 3) * B + bj) * 8u + (r & 7u)] = 0;
                }
                // ---------------- masked rows of this tile: 8 x 16 B per string, [M/8][B][8]; stored during the next walk ----------------
                {
                    const uint32_t cw[16] = {cq[0].x, cq[0].y, cq[0].z, cq[0].w, cq[1].x, cq[1].y, cq[1].z, cq[1].w,
                                             cq[2].x, cq[2].y, cq[2].z, cq[2].w, cq[3].x, cq[3].y, cq[3].z, cq[3].w};
                    const uint32_t mlo = (uint32_t)tm.mask, mhi = (uint32_t)(tm.mask >> 32);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const uint32_t mbyte = ((k < 4 ? mlo : mhi) >> (8 * (k & 3))) & 0xffu;
                        uint4 v = make_uint4(0, 0, 0, 0);
                        if (mbyte) {  // lib.rs:752-761
                            uint32_t o[8];
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const int p = k * 8 + i;
                                const uint32_t c = (cw[p >> 2] >> (8 * (p & 3))) & 0xffu;
                                const uint32_t sid = (sidq[p >> 2] >> (8 * (p & 3))) & 0xffu;
                                o[i] = ((mbyte >> i) & 1u) ? (c | (sid << 8)) : 0u;
                            }
                            v = make_uint4(o[0] | (o[1] << 16), o[2] | (o[3] << 16), o[4] | (o[5] << 16), o[6] | (o[7] << 16));
                        }
                        if (D == 1) pend[k] = v;  // leaves during the next tile's walk
                        else if (active && t0 + (uint32_t)k * 8u < M && !(a.debug & kDbgSkipMasked))
                            store16(mp + (size_t)k * mstep, v, false);  // D >= 2: the walk needs the registers; store now
                    }
                    pend_mp = mp;
                    mp += 8u * mstep;
                    have_pend = (D == 1);
                }
            }
            // the last tile's masked rows (only the octets that exist: [ceil(M/8)][B][8])
            if (active && have_pend && !(a.debug & kDbgSkipMasked)) {
                const uint32_t t0 = (ntiles - 1u) << 6;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (t0 + (uint32_t)k * 8u < M) *reinterpret_cast<uint4 *>(pend_mp + (size_t)k * mstep) = pend[k];
            }
            // ---------------- per-string status ----------------
            if (active) {
                accept = 0;
#pragma unroll
                for (int d = 0; d < D; ++d) accept |= (acc_state[d] == a.dc[d].accepted_state ? 1u : 0u) << d;
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
}

bool plan_witness_launch(WitnessArgs &a, int num_cus, LaunchInfo &out) {
    a.gs = 64;
    a.n_groups = (uint32_t)(((size_t)a.B + 63) / 64);
    out.gtab = 0;
    out.wide = 0;
    out.half = 0;
    // DFAs whose fused table leaves no room for the per-wave LDS areas are walked out of global memory (L2-resident)
    const size_t min_stage = (a.layout & 1u) ? (2 * 4096 + 4096 + 16) : wave_stage_bytes((int)a.D, 16);
    if (a.table_bytes + min_stage > kLdsLimit || (a.debug & kDbgForceGlobalTable)) out.gtab = 1;
    const uint32_t table_bytes_saved = a.table_bytes;
    struct Restore { WitnessArgs &a; uint32_t v; ~Restore() { a.table_bytes = v; } } restore{a, table_bytes_saved};
    if (out.gtab) a.table_bytes = 0;  // for the LDS budgeting below only; restored on return
    if ((a.layout & 1u) && a.half_image && ((out.gtab && !(a.debug & kDbgForceGlobalTable)) || (a.debug & kDbgForceHalf))) {
        // ---- loader/walker kernel on the HALF table (2-byte entries): DFAs of up to 256 states whose 4-byte table does not
        // fit LDS (cfg 5: 256 x 256 -> 128 KiB) stay LDS-resident instead of being walked out of L2 (kDbgForceHalf forces it)
        int pairs = 4;
        while (pairs > 1 && (size_t)a.n_groups < (size_t)num_cus * pairs) --pairs;
        for (; pairs >= 1; --pairs) {
            for (int ns = 4; ns >= 1; --ns) {
                const size_t lds = a.half_bytes + (size_t)pairs * (ns * 4096 + 16);
                if (lds > kLdsLimit) continue;
                out.split = 2; out.gtab = 0; out.wide = 0; out.half = 1;
                out.waves_per_wg = 2 * pairs;
                out.nslots = ns;
                out.lds_bytes = lds;
                const size_t need = ((size_t)a.n_groups + pairs - 1) / pairs;
                size_t per_cu = kLdsLimit / lds;
                if (per_cu * (size_t)(2 * pairs) > 8) per_cu = 8 / (size_t)(2 * pairs);
                if (per_cu < 1) per_cu = 1;
                const size_t cap = (size_t)num_cus * per_cu;
                out.grid = (int)(need < cap ? need : cap);
                if (out.grid < 1) out.grid = 1;
                return true;
            }
        }
    }
    if (a.layout & 1u) {
        // ---- loader/walker kernel: table + per pair a ring of up to 4 input tiles (4 KiB each)
        int pairs = 4;
        while (pairs > 1 && (size_t)a.n_groups < (size_t)num_cus * pairs) --pairs;
        for (; pairs >= 1; --pairs) {
            for (int ns = 4; ns >= 2; --ns) {
                const size_t lds = a.table_bytes + (size_t)pairs * (ns * 4096 + 4096 + 16);
                if (lds > kLdsLimit) continue;
                out.split = 2;
                // WIDE table: ~4x fewer instructions per row at D = 3; every D >= 2 batch takes it (same-box A/B
                // with spill-free kernels: 1.20 vs 1.33 ms at 262144 x 2048 B, D = 2; 3.48 vs 4.59 ms at 32768 x 32768 B, D = 3).
                // D = 1 gains nothing: its walk is LDS-latency-bound either way.
                out.wide = (a.wide_image && !out.gtab && !(a.debug & kDbgForceNarrow) && (a.D >= 2 || (a.debug & kDbgForceWide))) ? 1 : 0;
                out.waves_per_wg = 2 * pairs;
                out.nslots = ns;
                out.lds_bytes = lds;
                const size_t need = ((size_t)a.n_groups + pairs - 1) / pairs;
                size_t per_cu = kLdsLimit / lds;               // LDS
                if (per_cu * (size_t)(2 * pairs) > 8) per_cu = 8 / (size_t)(2 * pairs);  // 2 waves per SIMD (VGPRs)
                if (per_cu < 1) per_cu = 1;
                const size_t cap = (size_t)num_cus * per_cu;
                out.grid = (int)(need < cap ? need : cap);
                if (out.grid < 1) out.grid = 1;
                return true;
            }
        }
        return false;
    }
    // ---- walker/storer kernel: D in {1,2}, rows in multiples of 8, ring of >= 2 slots per pair
    if ((a.D == 1 || a.D == 2) && a.M % 8u == 0 && !(a.debug & kDbgForceOneWave) && !out.gtab) {
        const size_t slot = 64 * 128 + 64 * 8 + 64 * (a.D == 1 ? 32 : 16), fixed = 16 + 256;  // + the storer's LDS-DMA sink
        int pairs = 4;
        while (pairs > 1 && (size_t)a.n_groups < (size_t)num_cus * pairs) --pairs;  // small batches: spread over the CUs
        for (; pairs >= 1; --pairs) {
            if (a.table_bytes + pairs * (2 * slot + fixed) > kLdsLimit) continue;
            size_t ns = (kLdsLimit - a.table_bytes - pairs * fixed) / (pairs * slot);
            if (ns > 4) ns = 4;
            out.split = 1;
            out.waves_per_wg = 2 * pairs;
            out.nslots = (int)ns;
            out.lds_bytes = a.table_bytes + pairs * (ns * slot + fixed);
            const size_t need = ((size_t)a.n_groups + pairs - 1) / pairs;
            const size_t wgs_per_cu = kLdsLimit / out.lds_bytes < 1 ? 1 : kLdsLimit / out.lds_bytes;
            const size_t cap = (size_t)num_cus * (wgs_per_cu > 4 ? 4 : wgs_per_cu);
            out.grid = (int)(need < cap ? need : cap);
            if (out.grid < 1) out.grid = 1;
            return true;
        }
    }
    out.split = 0;
    out.nslots = 0;
    // group size: 64 strings per wave; smaller groups only for batches that would otherwise leave CUs without a wave
    // (two half-empty waves per SIMD were measured slower than one full one: the walk is issue-bound, not latency-bound)
    uint32_t gs = 64;
    if (a.debug & kDbgGroups32) gs = 32;
    while (gs > 16 && ((size_t)a.B + gs - 1) / gs < (size_t)num_cus) gs >>= 1;
    a.gs = gs;
    a.n_groups = (uint32_t)(((size_t)a.B + gs - 1) / gs);
    const size_t per_wave = wave_stage_bytes((int)a.D, gs);
    int waves = 0;
    for (int w = 8; w >= 1; --w) {
        if (a.table_bytes + per_wave * w <= kLdsLimit) { waves = w; break; }
    }
    if (!waves) return false;
    // fewer waves per workgroup when the batch cannot feed every CU otherwise
    while (waves > 1 && (size_t)a.n_groups < (size_t)num_cus * waves) --waves;
    out.waves_per_wg = waves;
    out.lds_bytes = a.table_bytes + per_wave * waves;
    const size_t wgs_per_cu = kLdsLimit / out.lds_bytes < 1 ? 1 : kLdsLimit / out.lds_bytes;
    const size_t max_wg = 8 / (size_t)waves < 1 ? 1 : 8 / (size_t)waves;  // <= 8 waves per CU: 2 per SIMD at 157 VGPRs
    const size_t need = ((size_t)a.n_groups + waves - 1) / waves;
    const size_t cap = (size_t)num_cus * (wgs_per_cu > max_wg ? max_wg : wgs_per_cu);
    out.grid = (int)(need < cap ? need : cap);
    if (out.grid < 1) out.grid = 1;
    return true;
}

// hipFuncSetAttribute costs several microseconds of host time: raise a kernel's dynamic-LDS limit only when a launch
// needs more than every earlier launch of that kernel did (the launch path is otherwise one hipLaunchKernelGGL).
template <class K>
static hipError_t ensure_lds(K k, std::atomic<size_t> &granted, size_t need) {
    if (need <= granted.load(std::memory_order_acquire)) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)need);
    if (e == hipSuccess) granted.store(need, std::memory_order_release);
    return e;
}

template <int D, int T>
static hipError_t launch_split(const WitnessArgs &a, const LaunchInfo &li, hipStream_t stream) {
    auto k = witness_split_kernel<D, T>;
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

template <int D, bool GTAB, bool WIDE = false, bool HALF = false>
static hipError_t launch_pm(const WitnessArgs &a, const LaunchInfo &li, hipStream_t stream) {
    auto k = witness_pm_kernel<D, GTAB, WIDE, HALF>;
    static std::atomic<size_t> granted[64];  // per device: the attribute is set on the current device's function
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipError_t e = ensure_lds(k, granted[dev & 63], li.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(li.grid), dim3(64 * li.waves_per_wg), li.lds_bytes, stream, a, (uint32_t)li.nslots);
    return hipGetLastError();
}

hipError_t launch_witness(const WitnessArgs &a, const LaunchInfo &li, hipStream_t stream) {
    if (li.split == 2) {
        if (li.half) return a.D == 1 ? launch_pm<1, false, false, true>(a, li, stream) : a.D == 2 ? launch_pm<2, false, false, true>(a, li, stream) : launch_pm<3, false, false, true>(a, li, stream);
        if (li.wide) return a.D == 1 ? launch_pm<1, false, true>(a, li, stream) : a.D == 2 ? launch_pm<2, false, true>(a, li, stream) : launch_pm<3, false, true>(a, li, stream);
        if (li.gtab) return a.D == 1 ? launch_pm<1, true>(a, li, stream) : a.D == 2 ? launch_pm<2, true>(a, li, stream) : launch_pm<3, true>(a, li, stream);
        return a.D == 1 ? launch_pm<1, false>(a, li, stream) : a.D == 2 ? launch_pm<2, false>(a, li, stream) : launch_pm<3, false>(a, li, stream);
    }
    if (li.split) return a.D == 1 ? launch_split<1, 32>(a, li, stream) : launch_split<2, 16>(a, li, stream);
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

// ---------------------------------------------------------------------------------------------
// states-in entry points (lib.rs:825-888): one thread per (def, row) looks the pair (s[i], s[i+1]) up.
// ---------------------------------------------------------------------------------------------
struct PairArgs {
    const uint64_t *states;
    uint64_t n;
    uint32_t D;
    const uint16_t *pt[3];
    uint32_t ns[3];
    uint16_t *tags;
};

__global__ void pair_tags_kernel(const PairArgs a) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n * a.D) return;
    const uint32_t d = (uint32_t)(i / a.n);
    const uint64_t r = i % a.n;
    const uint64_t cur = a.states[d * (a.n + 1) + r], next = a.states[d * (a.n + 1) + r + 1];
    uint16_t t = 0;
    if (cur < a.ns[d] && next < a.ns[d]) t = a.pt[d][cur * a.ns[d] + next];
    a.tags[i] = t;
}

hipError_t launch_pair_tags(const uint64_t *states, size_t n, uint32_t D, const uint16_t *const *pair_tags,
                            const uint32_t *n_states, uint16_t *tags, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    PairArgs a{};
    a.states = states; a.n = n; a.D = D; a.tags = tags;
    for (uint32_t d = 0; d < D; ++d) { a.pt[d] = pair_tags[d]; a.ns[d] = n_states[d]; }
    const uint64_t total = (uint64_t)n * D;
    hipLaunchKernelGGL(pair_tags_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

__global__ void endpoint_flags_kernel(const EndpointArgs a) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n * a.D) return;
    const uint32_t d = (uint32_t)(i / a.n);
    const uint64_t r = i % a.n;
    const uint64_t sid = a.substr_ids[i];
    uint8_t f = 0;
    if (sid != 0) {  // lib.rs:861-866, 874-879
        const uint64_t j = sid - a.id_offset[d];
        const uint64_t cur = a.states[d * (a.n + 1) + r], next = a.states[d * (a.n + 1) + r + 1];
        if (j < a.n_substrs[d]) {
            if (cur < a.n_states[d]) f |= a.member[d][j * a.n_states[d] + cur] & 1;
            if (next < a.n_states[d]) f |= a.member[d][j * a.n_states[d] + next] & 2;
        }
    }
    a.flags[i] = f;
}

hipError_t launch_endpoint_flags(const EndpointArgs &a, hipStream_t stream) {
    if (a.n == 0) return hipSuccess;
    const uint64_t total = a.n * a.D;
    hipLaunchKernelGGL(endpoint_flags_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// compact witness -> field cells.  A pair of threads per (string, row) expands the row's integers into what
// `Value::known(F::from(v))` holds for every advice cell the reference assigns (lib.rs:339-418, 473-519) and for its two
// result columns (lib.rs:752-771), F = bn256::Fr in Montgomery form (hrx_fr.h), column-major [col][string][row][4 limbs].
// Write-bound: 32 B per cell x (4 + 4 D) cells per row.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kFrRowsPerBlock = 512u;
__global__ __launch_bounds__(256) void fr_columns_kernel(const FrArgs a) {
    // thread = (row, half of the 32-byte cell): both lanes of a pair compute the cell, each stores its 16 bytes, so that a
    // store instruction writes 1 KiB of full lines (one lane per row stored half of every 32 bytes per instruction)
    const uint32_t half = threadIdx.x & 1u;
    const uint32_t bi = blockIdx.y;              // index inside the requested range
    const uint32_t b = a.b_begin + bi;
    const uint32_t n = min(a.lens[b], a.M);
    const bool pm = (a.layout & 1u) != 0, in_pm = (a.layout & 2u) != 0;
#pragma unroll 2
    for (uint32_t it = 0; it < kFrRowsPerBlock / 128u; ++it) {
    const uint32_t r = blockIdx.x * kFrRowsPerBlock + it * 128u + (threadIdx.x >> 1);
    if (r >= a.M) return;
    const uint32_t live = r < n ? 1u : 0u;
    uint32_t c = 0;
    if (live) c = in_pm ? a.chars[((size_t)(r >> 4) * a.B + b) * 16u + (r & 15u)] : a.chars[(size_t)b * a.stride + r];
    const size_t col_cells = (size_t)a.b_count * a.M;   // cells per column
    uint64_t *out = a.cells + ((size_t)bi * a.M + r) * 4u + half * 2u;
    auto put = [&](const uint32_t col, const uint32_t v) {
        uint32_t w[8];
        fr_from_u32(v, w, a.canonical != 0);
        uint4 *p = reinterpret_cast<uint4 *>(out + (size_t)col * col_cells * 4u);
        *p = half ? make_uint4(w[4], w[5], w[6], w[7]) : make_uint4(w[0], w[1], w[2], w[3]);
    };
    put(0, live);   // char_enable                        lib.rs:342,346
    put(1, c);      // characters                         lib.rs:343,347
    for (uint32_t d = 0; d < a.D; ++d) {
        const uint32_t rec = pm ? a.records[(((size_t)(r >> 2) * a.D + d) * a.B + b) * 4u + (r & 3u)]
                                : a.records[((size_t)b * a.rec_pitch + r) * a.D + d];
        put(2 + 4 * d, rec & 0xffffu);            // states[d]         lib.rs:390,415
        put(3 + 4 * d, (rec >> 16) & 0xffu);      // substr_ids[d]     lib.rs:394,405
        put(4 + 4 * d, (rec >> 24) & 1u);         // start_enable[d]   lib.rs:483-491
        put(5 + 4 * d, (rec >> 25) & 1u);         // end_enable[d]     lib.rs:502-511
    }
    const uint32_t mk = pm ? a.masked[((size_t)(r >> 3) * a.B + b) * 8u + (r & 7u)] : a.masked[(size_t)b * a.msk_pitch + r];
    put(2 + 4 * a.D, mk & 0xffu);                 // masked_characters  lib.rs:752-757
    put(3 + 4 * a.D, mk >> 8);                    // all_substr_ids     lib.rs:758-761
    }
}

hipError_t launch_fr_columns(const FrArgs &a, hipStream_t stream) {
    if (a.b_count == 0 || a.M == 0) return hipSuccess;
    hipLaunchKernelGGL(fr_columns_kernel, dim3((a.M + kFrRowsPerBlock - 1u) / kFrRowsPerBlock, a.b_count), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace hrx
