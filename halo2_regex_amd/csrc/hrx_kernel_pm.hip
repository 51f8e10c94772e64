// hrx_kernel_pm.hip — gfx950 kernel for the POSITION-MAJOR buffers (the headline path).
#include <hip/hip_runtime.h>

#include "hrx_device.h"

namespace hrx {

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
template <int D, bool SM = false>
struct GlobalSink {
    static constexpr bool kSidq = true;
    unsigned char *rp;
    size_t plane, rstep;
    bool do_store, nt_rec, nt_msk;
    const uint4 (&pend)[8];
    unsigned char *pend_mp;
    size_t mstep;
    bool pend_store;
    uint4 held[SM ? D : 1];   // SM: the quads of defs 0..D-2, until the last def's arrives
    __device__ __forceinline__ void quad(const int d, const int p, const bool full, const int mrem, const uint4 &v) {
        if (SM) {
            // string-major records [B][pitch][D]: four rows of this string are 16*D contiguous bytes, rows outermost — the
            // lane writes them itself (16-byte pieces that L2 merges into lines), no LDS transpose.  M % 4 == 0 here.
            // (At D = 1 / 2 the walker/storer kernel's LDS transpose is 2x faster than this: 100 vs 236 us, 1.29 vs 2.31 ms.)
            held[d] = v;
            if (d == D - 1) {
                uint32_t w[4 * D];
#pragma unroll
                for (int dd = 0; dd < D; ++dd) {
                    w[0 * D + dd] = held[dd].x; w[1 * D + dd] = held[dd].y; w[2 * D + dd] = held[dd].z; w[3 * D + dd] = held[dd].w;
                }
                if (do_store && (full || (p & ~3) <= mrem)) {
#pragma unroll
                    for (int k = 0; k < D; ++k) store16(rp + 16 * k, make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]), nt_rec);
                }
                rp += rstep;
            }
            return;
        }
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

template <int D, bool GTAB, bool WIDE, bool HALF = false, bool SM = false>
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
            // SM (string-major outputs from this kernel: D = 3, which the walker/storer kernel's 128-byte string-tiles do not
            // cover): records [B][pitch][D], masked [B][pitch] — same walk, the lane's own strides
            unsigned char *rp = SM ? reinterpret_cast<unsigned char *>(a.records) + (size_t)bc * a.rec_pitch * D * 4u
                                   : reinterpret_cast<unsigned char *>(a.records) + (size_t)bc * 16u * ((a.debug & kDbgInterleavedDefs) ? D : 1);
            // (kDbgFixedLines, profiling only: every quad / octet of a string lands on the first one — same store instructions, no new lines or pages)
            const size_t rstep = (a.debug & kDbgFixedLines) ? (size_t)0 : SM ? (size_t)16u * D : (size_t)B * 16u * D;  // one quad of rows further: [M/4][D][B][4]
            unsigned char *mp = SM ? reinterpret_cast<unsigned char *>(a.masked) + (size_t)bc * a.msk_pitch * 2u : reinterpret_cast<unsigned char *>(a.masked) + (size_t)bc * 16u;
            const size_t mstep = (a.debug & kDbgFixedLines) ? (size_t)0 : SM ? (size_t)16u : (size_t)B * 16u;      // 8 rows further: [M/8][B][8]
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
                GlobalSink<D, SM> sink{rp, (a.debug & kDbgInterleavedDefs) ? (size_t)16u : (size_t)B * 16u, rstep, do_store, (a.debug & kDbgNtRecords) != 0, (a.debug & kDbgNtMasked) != 0,
                                       pend, pend_mp, mstep, pend_store, {}};
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
                        a.masked[SM ? (size_t)bj * a.msk_pitch + r : ((size_t)(r >> 3) * B + bj) * 8u + (r & 7u)] = 0;
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

template <int D, bool GTAB, bool WIDE = false, bool HALF = false, bool SM = false>
static hipError_t launch_pm(const WitnessArgs &a, const LaunchInfo &li, hipStream_t stream) {
    auto k = witness_pm_kernel<D, GTAB, WIDE, HALF, SM>;
    static std::atomic<size_t> granted[64];  // per device: the attribute is set on the current device's function
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipError_t e = ensure_lds(k, granted[dev & 63], li.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(li.grid), dim3(64 * li.waves_per_wg), li.lds_bytes, stream, a, (uint32_t)li.nslots);
    return hipGetLastError();
}

hipError_t launch_witness_pm(const WitnessArgs &a, const LaunchInfo &li, hipStream_t stream) {
    if (!(a.layout & 1u)) {   // string-major outputs, D = 3 (plan_witness_launch)
        if (a.D != 3 || li.half || li.gtab) return hipErrorInvalidValue;
        return li.wide ? launch_pm<3, false, true, false, true>(a, li, stream) : launch_pm<3, false, false, false, true>(a, li, stream);
    }
    if (li.half) return a.D == 1 ? launch_pm<1, false, false, true>(a, li, stream) : a.D == 2 ? launch_pm<2, false, false, true>(a, li, stream) : launch_pm<3, false, false, true>(a, li, stream);
    if (li.wide) return a.D == 1 ? launch_pm<1, false, true>(a, li, stream) : a.D == 2 ? launch_pm<2, false, true>(a, li, stream) : launch_pm<3, false, true>(a, li, stream);
    if (li.gtab) return a.D == 1 ? launch_pm<1, true>(a, li, stream) : a.D == 2 ? launch_pm<2, true>(a, li, stream) : launch_pm<3, true>(a, li, stream);
    return a.D == 1 ? launch_pm<1, false>(a, li, stream) : a.D == 2 ? launch_pm<2, false>(a, li, stream) : launch_pm<3, false>(a, li, stream);
}

}  // namespace hrx
