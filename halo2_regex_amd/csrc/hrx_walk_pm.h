// hrx_walk_pm.h — the per-tile walk of the position-major kernel (hrx_kernel_pm.hip): sink policy,
// the narrow / HALF walk (walk_tile_pm) and the WIDE walk (walk_tile_pm_wide).  Device-only.
#pragma once
#include "hrx_device.h"

namespace hrx {

constexpr uint32_t kPmTileBytes = 64u * 64u;  // 64 strings x 64 input bytes per tile

// 16 bytes per lane, 1 KiB of full lines per wave.  nt: a non-temporal (streaming) store.  The nt form is inline asm on
// purpose: with `if (nt) __builtin_nontemporal_store(..) else *p = v` LLVM merges the two stores of the diamond into ONE
// ordinary store and drops the hint (round 1's "nt changes nothing" A/B measured exactly that: no store in the code object
// carried the nt bit).  The trailing s_nop 1 covers the TWO wait states a >64-bit VMEM store needs on gfx940+ before a VALU may
// overwrite its data registers (one on older parts); the hazard recogniser does not look into inline asm.  With s_nop 0 the
// def-parallel kernel, which re-initialises the same four registers right after each masked-row store, lost the first dword
// of some octets (rows 32-33 of a tile, nondeterministically).
__device__ __forceinline__ void store16(unsigned char *p, const uint4 &v, const bool nt) {
    if (nt) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(p), "v"(v4u32{v.x, v.y, v.z, v.w}));
    else *reinterpret_cast<uint4 *>(p) = v;
}

// One octet of masked rows (lib.rs:752-761): rows 8k .. 8k+7 of a string as 8 x {masked_char, masked_substr_id} = 16 bytes.
// c0 / c1: the octet's raw bytes (4 rows per dword), s0 / s1: its substr-id sums (one byte per row), mbyte: its 8 mask bits.
// Per four rows: the nibble of mask bits is spread to one bit per byte (n * 0x204081 puts bit i at bit 8 i: the four partial
// products land on disjoint bit positions, so nothing carries), widened to 0x00 / 0xff bytes, ANDed onto both sources, and
// two v_perm_b32 interleave the surviving bytes into u16 {char, id} pairs — 9 VALU per four rows where the row-by-row
// select took ~6 per row (the tile-end work is serial to the dependent chain of the walk, so it is paid in full).
__device__ __forceinline__ uint4 masked_octet(const uint32_t c0, const uint32_t c1, const uint32_t s0, const uint32_t s1, const uint32_t mbyte) {
    auto half = [](const uint32_t c, const uint32_t s, const uint32_t nib, uint32_t &lo, uint32_t &hi) {
        const uint32_t x = __umul24(nib, 0x204081u) & 0x01010101u;
        const uint32_t bm = (x << 8) - x;
        const uint32_t cm = c & bm, sm = s & bm;
        lo = __builtin_amdgcn_perm(sm, cm, 0x05010400u);   // {c.b0, s.b0, c.b1, s.b1}
        hi = __builtin_amdgcn_perm(sm, cm, 0x07030602u);   // {c.b2, s.b2, c.b3, s.b3}
    };
    uint4 v;
    half(c0, s0, mbyte & 0xfu, v.x, v.y);
    half(c1, s1, mbyte >> 4, v.z, v.w);
    return v;
}

// Where a walker's finished rows go: straight to memory from its registers.  quad(d, p, ..) stores four rows of def d
// (16 B per lane, 1 KiB contiguous per wave) into its plane of [ceil(M/4)][D][B][4]; row(p) lets the previous tile's masked
// rows leave one 16-byte piece every 8 rows.  (The walk functions take the sink as a policy: a variant that handed the
// rows to a third "storer" wave through an LDS out-ring, so that the walker issued no vector-memory instruction at all, was
// built and measured in round 1 — every global store does cost the issuing wave 75-125 cycles, but the ds_write_b128 +
// hand-over cost the walker as much, and where all walker slots are busy the launch is bound by the memory system's mixed
// read/write rate anyway: 97 vs 89 us on the headline workload, 3.41 vs 3.39 ms on cfg 4.  Dropped; NOTES_MEASUREMENTS.md §4.)
template <int D, bool SM = false>
struct GlobalSink {
    static constexpr bool kSidq = true;
    unsigned char *rp;
    const size_t (&poff)[D];   // def d's plane relative to def 0's: d * nb * 16 in the interleaved [M/4][D][nb][4], the distance of the buffers with WitnessArgs::rec_planes
    size_t rstep;              // to the next quad of rows (D == 1: after an ODD quad of the tile; the tile has 16)
    size_t rstep_even, soff1;  // D == 1: the step after an EVEN quad (= rstep, or 0 with two row stripes) and the second stripe relative to the first (0 without: WitnessArgs::rec_stripes)
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
        if (D == 1) {   // (p is a compile-time constant in the unrolled walks: the quad's parity picks the scalars)
            const bool odd = ((p >> 2) & 1) != 0;
            if (do_store && (full || (p & ~3) <= mrem)) store16(rp + (odd ? soff1 : (size_t)0), v, nt_rec);
            rp += odd ? rstep : rstep_even;
            return;
        }
        if (do_store && (full || (p & ~3) <= mrem)) store16(rp + poff[d], v, nt_rec);
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
// BYTE-table walk (hrx_lane.h; one def).  Per row two LDS reads, neither of which waits for the other:
//   iteration p:  next-state byte of row p  (the dependent chain: address = state << 8 | byte, ONE v_perm_b32),
//                 pair slot of row p - 1    (address ((state * A4 + next * B4) & (slots - 1) * 4) | ptab_off, both known since the previous iteration),
//   and in their shadow the record of row p - 2, whose slot arrived an iteration ago.
// A wave alone issues one vector instruction per 4 cycles (MI355X_MICROARCH.md: 'vector-instruction ISSUE cost'), and round 3's walker
// spent 27 of them per row (disassembly: 108 issue cycles against the ~55 of the chain itself; in-kernel stamps: 164 cycles per row,
// the launch's whole duration) on work that does not need the chain: the three tile bitvectors and the substr-id bytes, one row at a
// time.  Now the 4-byte pair slot carries the record's high half ready-made (record = slot & 0xffff0000 | state: one v_and_or_b32) and a
// TAG BYTE (substr id | is_start << 6 | is_end << 7) that the walker only packs four to a dword (one v_perm_b32 per row): the finisher
// wave derives bitvectors and id bytes from the 16 dwords of a tile with byte-parallel arithmetic (hrx_kernel_pm.hip byte_tile_bits:
// ~3 instructions per row instead of ~12).  10 vector instructions per row are left here.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) const uint8_t lds_cu8;
__device__ __forceinline__ uint32_t lds_u8(uint32_t off) { return *(lds_cu8 *)(uintptr_t)off; }

template <bool FULL, class Sink>
__device__ __forceinline__ void walk_tile_pm_byte(LaneRegs<1> &L, const uint4 (&cq)[4], const WitnessArgs &a, Sink &sink, int rem, int mrem,
                                                  uint32_t (&tagq)[16], uint32_t (&acc_state)[1]) {
    uint32_t rbuf[4];
    const uint32_t cw[16] = {cq[0].x, cq[0].y, cq[0].z, cq[0].w, cq[1].x, cq[1].y, cq[1].z, cq[1].w,
                             cq[2].x, cq[2].y, cq[2].z, cq[2].w, cq[3].x, cq[3].y, cq[3].z, cq[3].w};
    const uint32_t A4 = a.byte_mul_a4, B4 = a.byte_mul_b4, smask = a.byte_slot_mask4;
    uint32_t ptab = a.byte_ptab_off;
    asm volatile("" : "+v"(ptab));          // in a vector register: (hash & smask) | ptab is ONE v_and_or_b32 then (an instruction reads at most one scalar register)
    uint32_t cur = L.e[0];                  // state at the row whose chain lookup is issued next
    uint32_t s1 = 0, n1 = 0;                // row p - 1: its state and its next state
    uint32_t s2 = 0, n2 = 0, pe2 = 0;       // row p - 2: its state, its next state and its pair slot

    auto post = [&](const int p, uint32_t state, const uint32_t next, const uint32_t pe) {
        uint32_t x = (pe & 0xffu) == next ? pe : 0u;                 // the slot's key is this pair's next state: (state, next) is tagged
        if (!FULL) {
            if (p >= rem) x = 0;                                     // padding rows: their lookups ran on stand-in states (lib.rs:404-418)
            if (p > rem) state = a.dc[0].dummy_state;                // lib.rs:413
            if (p >= mrem) x &= ~(kRecEndBit | 0x8000u);             // end_enable of row M - 1 is never assigned (lib.rs:501): record half and tag byte
            if (p == rem) acc_state[0] = state;                      // the state at row n (lib.rs:437-457)
        }
        rbuf[p & 3] = (x & 0xffff0000u) | state;
        if ((p & 3) == 3) sink.quad(0, p, FULL, mrem, make_uint4(rbuf[0], rbuf[1], rbuf[2], rbuf[3]));
        // the row's tag byte (byte 1 of the slot) -> byte p & 3 of the quad's dword
        uint32_t &t = tagq[p >> 2];
        if ((p & 3) == 0) t = (x >> 8) & 0xffu;
        else if ((p & 3) == 1) t = __builtin_amdgcn_perm(x, t, 0x0c0c0500u);
        else if ((p & 3) == 2) t = __builtin_amdgcn_perm(x, t, 0x0c050100u);
        else t = __builtin_amdgcn_perm(x, t, 0x05020100u);
        sink.row(p);
    };
#pragma unroll
    for (int p = 0; p < 66; ++p) {
        uint32_t raw_n = 0, raw_pe = 0;
        if (p < 64)   // delta(state, byte): lib.rs:810.  address = state << 8 | byte (p & 3) of the quad's dword
            raw_n = lds_u8(__builtin_amdgcn_perm(cur, cw[p >> 2], 0x0c0c0400u | (uint32_t)(p & 3)));
        if (p >= 1 && p < 65) raw_pe = lds_u32(((__umul24(s1, A4) + __umul24(n1, B4)) & smask) | ptab);
        if (p >= 2) {
            post(p - 2, s2, n2, pe2);
            asm volatile("" : "+v"(tagq[(p - 2) >> 2]));
            if (!FULL) asm volatile("" : "+v"(acc_state[0]));
        }
        __builtin_amdgcn_sched_barrier(0);
        s2 = s1; n2 = n1; pe2 = raw_pe;
        if (p < 64) {
            const uint32_t nxt = (FULL || p < rem) ? raw_n : 0u;    // rows >= n: any valid row; post() writes the dummy state and no tag
            L.mx[0] = max(L.mx[0], nxt);                              // reaching the dead row = an undefined transition (lib.rs:817)
            asm volatile("" : "+v"(L.mx[0]));
            s1 = cur; n1 = nxt;
            cur = nxt;
        }
    }
    L.e[0] = cur;
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

// RS / CLS: the CLASS-WIDE table of the def-parallel kernel over more than three defs (hrx_lane.h kCwRowShift): the row field sits at bit RS, and `cq` holds the bytes'
// columns already (class x 8, looked up in the def's LUT by the caller) instead of the bytes.
template <int D, bool FULL, class Sink, int RS = kWideRowShift, bool CLS = false>
__device__ __forceinline__ TileBits walk_tile_pm_wide(LaneRegs<D> &L, const uint4 (&cq)[4], const WitnessArgs &a, Sink &sink, int rem, int mrem,
                                                      uint32_t &tile_ov, uint32_t (&sidq)[16], uint32_t (&acc_state)[D]) {
    constexpr uint32_t kRowField = RS == kWideRowShift ? 0xffu : 0x3ffu, kRowMaskT = kRowField << RS, kByteMask = CLS ? 0xffffffffu : 0x7f7f7f7fu;
    uint32_t st[2] = {0, 0}, en1[2] = {0, 0}, ch[2] = {0, 0};
    uint32_t rbuf[D][4];
    uint32_t ov = 0;
    // bytes >= 128 have no column: they are masked here and the tile is re-walked by the caller
    const uint32_t cw[16] = {cq[0].x & kByteMask, cq[0].y & kByteMask, cq[0].z & kByteMask, cq[0].w & kByteMask,
                             cq[1].x & kByteMask, cq[1].y & kByteMask, cq[1].z & kByteMask, cq[1].w & kByteMask,
                             cq[2].x & kByteMask, cq[2].y & kByteMask, cq[2].z & kByteMask, cq[2].w & kByteMask,
                             cq[3].x & kByteMask, cq[3].y & kByteMask, cq[3].z & kByteMask, cq[3].w & kByteMask};
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
            const uint32_t c8 = CLS ? ((cw[q] >> (8 * k)) & 0xffu) : ((cw[q] >> (8 * k)) & 0xffu) << 3;
            uint2 raw[D];
#pragma unroll
            for (int d = 0; d < D; ++d) raw[d] = lds_u64((lo[d] & kRowMaskT) | c8);   // delta(state, byte): lib.rs:810
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
                    const uint32_t state_here = ((prev >> RS) & kRowField) - a.dc[d].row_base;
                    if (p == rem) acc_state[d] = state_here;                       // the state at row n (lib.rs:437-457)
                    lo[d] = live ? raw[d].x : a.dc[d].dummy_entry;                 // rows >= n: lib.rs:404-418
                    phi[d] = live ? raw[d].y : (p == rem ? state_here : (a.dc[d].dummy_entry >> RS) - a.dc[d].row_base);
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

}  // namespace hrx
