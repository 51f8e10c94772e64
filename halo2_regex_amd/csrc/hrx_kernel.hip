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

#include "hrx_kernel.hpp"
#include "hrx_lane.h"

namespace hrx {

extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

// The kernels declare no static LDS, so the dynamic segment starts at LDS address 0 and a byte offset IS the
// LDS address: reads go through integer->address_space(3) casts so that no base add sits on the walk's
// dependent chain (witness_kernel traps if the assumption ever breaks).
typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const v4u32 lds_cv4u32;
__device__ __forceinline__ uint32_t lds_u32(uint32_t off) { return *(lds_cu32 *)(uintptr_t)off; }
__device__ __forceinline__ uint4 lds_u128(uint32_t off) {
    const v4u32 v = *(lds_cv4u32 *)(uintptr_t)off;
    return make_uint4(v.x, v.y, v.z, v.w);
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
template <int D, bool FULL>
__device__ __forceinline__ TileBits walk_tile(LaneRegs<D> &L, const uint4 (&cq)[4], const WitnessArgs &a,
                                              uint32_t my_rec, int rem, int mrem, uint32_t t0) {
    uint32_t st[2] = {0, 0}, en1[2] = {0, 0}, ch[2] = {0, 0};
    uint32_t rbuf[4];
    const uint32_t cw[16] = {cq[0].x, cq[0].y, cq[0].z, cq[0].w, cq[1].x, cq[1].y, cq[1].z, cq[1].w,
                             cq[2].x, cq[2].y, cq[2].z, cq[2].w, cq[3].x, cq[3].y, cq[3].z, cq[3].w};
#pragma unroll
    for (int q = 0; q < 16; ++q) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = q * 4 + k;
            const uint32_t c4 = ((cw[q] >> (8 * k)) & 0xffu) << 2;
            const bool live = FULL ? true : (p < rem);
            uint32_t sid = 0, stn = 0, enn = 0;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const uint32_t state = (L.e[d] >> kNextShift) - (d ? a.dc[d].row_base : 0u);
                uint32_t ent = lds_u32((L.e[d] & ~kTagMask) | c4);  // delta(state, byte): lib.rs:810
                if (!FULL) ent = live ? ent : a.dc[d].dummy_entry;
                uint32_t tag = ent & kTagMask;
                if (!FULL) {
                    if (p >= mrem) tag &= ~kTagEnd;
                }
                const int slot = (p * D + d) & 3;
                rbuf[slot] = state | (tag << 16);
                if (slot == 3)
                    *reinterpret_cast<uint4 *>(smem + my_rec + (p * D + d - 3) * 4) =
                        make_uint4(rbuf[0], rbuf[1], rbuf[2], rbuf[3]);
                L.e[d] = ent;
                L.mx[d] = max(L.mx[d], ent);
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
        }
    }
    TileBits tb;
    tb.st = (uint64_t)st[0] | ((uint64_t)st[1] << 32);
    tb.en1 = (uint64_t)en1[0] | ((uint64_t)en1[1] << 32);
    tb.ch = (uint64_t)ch[0] | ((uint64_t)ch[1] << 32);
    return tb;
}

__device__ __forceinline__ void load_chars(uint4 (&q)[4], const uint8_t *cptr, uint32_t t0, uint32_t n, bool active) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t off = t0 + 16u * i;
        q[i] = (active && off < n) ? *reinterpret_cast<const uint4 *>(cptr + off) : make_uint4(0, 0, 0, 0);
    }
}

// D: number of RegexDefs.  ALIGNED: M % 8 == 0, so every string-tile of records and masked rows starts on a
// 16-byte boundary and the store phase moves 16 B per lane.
template <int D, bool ALIGNED>
__global__ __launch_bounds__(256) void witness_kernel(const WitnessArgs a) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t waves = blockDim.x >> 6;
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();

    // ---- stage the fused tables of all defs into LDS (offset 0) ----
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.table_image);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = threadIdx.x; i < a.table_bytes / 16u; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();

    constexpr uint32_t RSB = 256u * D + 16u;  // bytes per staged string-tile of records (+16 B: bank spread)
    constexpr uint32_t CSB = 80u;             // bytes per staged string-tile of chars
    const uint32_t rec_base = a.table_bytes + wave * (uint32_t)wave_stage_bytes(D);
    const uint32_t chr_base = rec_base + 64u * RSB;
    const uint32_t mb_base = chr_base + 64u * CSB;
    const uint32_t my_rec = rec_base + lane * RSB;
    const uint32_t M = a.M;
    const uint32_t ntiles = (M + 63u) >> 6;

    for (uint32_t g = blockIdx.x * waves + wave; g < a.n_groups; g += gridDim.x * waves) {
        const uint32_t b0 = g * 64u;
        const uint32_t b = b0 + lane;
        const bool active = b < a.B;
        const uint32_t n_raw = active ? a.lens[b] : M;
        const bool badlen = n_raw > M;
        const uint32_t n = badlen ? M : n_raw;
        const uint32_t min_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_min_u32(n));
        const uint8_t *cptr = a.chars + (size_t)b * a.stride;

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
        load_chars(cq, cptr, 0, n, active);

        for (uint32_t t = 0; t < ntiles; ++t) {
            const uint32_t t0 = t << 6;
            if (t + 1 < ntiles) load_chars(nq, cptr, t0 + 64u, n, active);  // prefetch the next tile's bytes

            // ---------------- walk + tag: lib.rs:804-888 ----------------
            TileBits tb;
            const bool full = (t0 + 64u < min_n);
            if (full)
                tb = walk_tile<D, true>(L, cq, a, my_rec, 0, 0, t0);
            else
                tb = walk_tile<D, false>(L, cq, a, my_rec, (int)n - (int)t0, (int)M - 1 - (int)t0, t0);

            bool chars_staged = false;
            auto stage_chars = [&]() {
                if (!chars_staged) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        *reinterpret_cast<uint4 *>(smem + chr_base + lane * CSB + 16u * i) = cq[i];
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
                                err_char[d] = smem[chr_base + lane * CSB + p];
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
            *reinterpret_cast<uint64_t *>(smem + mb_base + lane * 8u) = tm.mask;
            const bool any_mask = __any(tm.mask != 0);
            if (any_mask) stage_chars();

            // an earlier optimistic end_mask = 1 turned out wrong: zero those masked rows (rare)
            uint64_t fixm = __ballot(tm.fix != 0);
            if (fixm) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                while (fixm) {
                    const int j = __ffsll((unsigned long long)fixm) - 1;
                    fixm &= fixm - 1;
                    const uint32_t fs = (uint32_t)__builtin_amdgcn_readlane((int)tm.fix_start, j);
                    uint16_t *mrow = a.masked + (size_t)(b0 + j) * M;
                    for (uint32_t r = (fs & ~63u) + lane; r < t0; r += 64u)
                        if (r >= fs) mrow[r] = 0;
                }
            }

            // ---------------- store phase: LDS-transposed, coalesced ----------------
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (ALIGNED) {
                // records: a string-tile is 256*D contiguous bytes = 16*D chunks of 16 B
                constexpr uint32_t CPS = 16u * D;
                const uint32_t lim = (M - t0 >= 64u ? 64u : M - t0) * D / 4u;  // valid chunks per string-tile
#pragma unroll 4
                for (uint32_t it = 0; it < CPS; ++it) {
                    const uint32_t chunk = it * 64u + lane;
                    const uint32_t js = chunk / CPS, w = chunk % CPS;
                    if (b0 + js < a.B && w < lim) {
                        const uint4 v = lds_u128(rec_base + js * RSB + w * 16u);
                        uint32_t *dst = a.records + ((size_t)(b0 + js) * M + t0) * D + w * 4u;
                        *reinterpret_cast<uint4 *>(dst) = v;
                    }
                }
                // masked rows: a string-tile is 128 contiguous bytes = 8 chunks of 16 B (8 rows each)
                const uint32_t mlim = (M - t0 >= 64u ? 64u : M - t0) / 8u;
#pragma unroll 2
                for (uint32_t it = 0; it < 8u; ++it) {
                    const uint32_t js = it * 8u + (lane >> 3), w = lane & 7u;
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (any_mask) {
                        const uint32_t mbyte = smem[mb_base + js * 8u + w];
                        if (mbyte) {
                            const uint2 cc = *reinterpret_cast<const uint2 *>(smem + chr_base + js * CSB + w * 8u);
                            uint32_t o[8];
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                uint32_t sid = 0;
#pragma unroll
                                for (int d = 0; d < D; ++d)
                                    sid += (lds_u32(rec_base + js * RSB + ((w * 8u + i) * D + d) * 4u) >> 16) & 0xffu;
                                const uint32_t c = ((i < 4 ? cc.x : cc.y) >> (8 * (i & 3))) & 0xffu;
                                o[i] = ((mbyte >> i) & 1u) ? (c | (sid << 8)) : 0u;  // lib.rs:752-761
                            }
                            v = make_uint4(o[0] | (o[1] << 16), o[2] | (o[3] << 16), o[4] | (o[5] << 16), o[6] | (o[7] << 16));
                        }
                    }
                    if (b0 + js < a.B && w < mlim)
                        *reinterpret_cast<uint4 *>(a.masked + (size_t)(b0 + js) * M + t0 + w * 8u) = v;
                }
            } else {
                // generic M: one dword / one u16 per lane, still contiguous per string
                const uint32_t rows = (M - t0 >= 64u ? 64u : M - t0);
                if (any_mask) stage_chars();
                for (uint32_t js = 0; js < 64u && b0 + js < a.B; ++js) {
#pragma unroll
                    for (int dd = 0; dd < D; ++dd) {
                        const uint32_t i = dd * 64u + lane;
                        if (i < rows * D)
                            a.records[((size_t)(b0 + js) * M + t0) * D + i] = lds_u32(rec_base + js * RSB + i * 4u);
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
                        a.masked[(size_t)(b0 + js) * M + t0 + lane] = (uint16_t)o;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();

#pragma unroll
            for (int i = 0; i < 4; ++i) cq[i] = nq[i];
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

bool plan_witness_launch(const WitnessArgs &a, int num_cus, LaunchInfo &out) {
    const size_t per_wave = wave_stage_bytes((int)a.D);
    int waves = 0;
    for (int w = 4; w >= 1; --w) {
        if (a.table_bytes + per_wave * w <= kLdsLimit) { waves = w; break; }
    }
    if (!waves) return false;
    // fewer waves per workgroup when the batch cannot feed every CU otherwise
    while (waves > 1 && (size_t)a.n_groups < (size_t)num_cus * waves) --waves;
    out.waves_per_wg = waves;
    out.lds_bytes = a.table_bytes + per_wave * waves;
    const int wgs_per_cu = (int)(kLdsLimit / out.lds_bytes) < 1 ? 1 : (int)(kLdsLimit / out.lds_bytes);
    const size_t need = ((size_t)a.n_groups + waves - 1) / waves;
    const size_t cap = (size_t)num_cus * (size_t)(wgs_per_cu > 8 ? 8 : wgs_per_cu);
    out.grid = (int)(need < cap ? need : cap);
    if (out.grid < 1) out.grid = 1;
    return true;
}

template <int D, bool ALIGNED>
static hipError_t launch_t(const WitnessArgs &a, const LaunchInfo &li, hipStream_t stream) {
    auto k = witness_kernel<D, ALIGNED>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)li.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(li.grid), dim3(64 * li.waves_per_wg), li.lds_bytes, stream, a);
    return hipGetLastError();
}

hipError_t launch_witness(const WitnessArgs &a, const LaunchInfo &li, hipStream_t stream) {
    const bool al = (a.M % 8u) == 0;
    switch (a.D) {
        case 1: return al ? launch_t<1, true>(a, li, stream) : launch_t<1, false>(a, li, stream);
        case 2: return al ? launch_t<2, true>(a, li, stream) : launch_t<2, false>(a, li, stream);
        case 3: return al ? launch_t<3, true>(a, li, stream) : launch_t<3, false>(a, li, stream);
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

}  // namespace hrx
