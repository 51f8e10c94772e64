// hrx_lane.h — per-string ("per-lane") integer logic of the witness generator.
//
// One GPU lane owns one input string.  Everything a lane needs to know about a
// 64-row tile of its string is kept as 64-bit position bitvectors in its own
// registers (bit p <-> witness row t0+p), so the reference's two sequential
// "last event wins" mask scans (src/lib.rs:598-645 forward, 663-714 backward)
// become a handful of 64-bit adds per tile instead of ~24 field-gate calls per
// row.  The header is plain C++ so that the same code is compiled by hipcc for
// the kernel and by g++ for the CPU algorithm test (tests/sim); it is not a
// CPU fallback — the product path only ever runs it on the device.
#pragma once
#include <stdint.h>

#if defined(__HIP__)  // clang in HIP mode (hipcc), host and device passes alike
#define HRX_HD __attribute__((host)) __attribute__((device)) inline __attribute__((always_inline))
#else
#define HRX_HD inline
#endif

namespace hrx {

constexpr int kTile = 64;  // witness rows per tile = width of the position bitvectors

// ---------------------------------------------------------------------------
// Fused dense-table entry: one u32 per (state, byte).
//   bits 10..31  next_state * 1024  = byte offset of the next state's 256-entry row
//   bits  0.. 7  substr_id of the transition (state,next)   table.rs:110-120 / lib.rs:831-840
//   bit   8      is_start: substr_id != 0 && state in start_states     lib.rs:861-866
//   bit   9      is_end of the NEXT row: substr_id != 0 && next in end_states   lib.rs:874-879
// so  (entry & ~0x3ff) + 4*byte  is the LDS address of the next lookup (one v_and_or),
// and (entry & 0x3ff) << 16 | state is the compact witness record of the row.
// ---------------------------------------------------------------------------
constexpr uint32_t kTagMask = 0x3ffu;
constexpr int kNextShift = 10;
constexpr uint32_t kTagStart = 1u << 8;
constexpr uint32_t kTagEnd = 1u << 9;

// WIDE table (position-major kernel, DFAs whose symbols are all < 128): 8-byte entries, 128 columns, the same 1 KiB
// per state row.  lo word = the walk's chain word, hi word = the finished compact record of the row.
//   lo: bits 3..9 zero (column), 10..17 absolute table row of the NEXT state (the LDS byte address of that row),
//       18..19 zero (carry guard), 20..27 substr_id, 28..29 is_start (a 2-bit counter field), 30..31 is_end (ditto):
//       the per-row sums over the defs are ONE v_add3_u32 of the lo words, no masking.
//   hi: state | substr_id << 16 | is_start << 24 | is_end << 25  (SURVEY App. A.4 record).
constexpr int kWideRowShift = 10;
constexpr uint32_t kWideRowMask = 0xffu << kWideRowShift;
constexpr int kWideSidShift = 20, kWideStartShift = 28, kWideEndShift = 30;

// CLASS-WIDE table (def-parallel kernel over more than three defs): the WIDE entry over byte classes — 32 columns x 8 B = 256 B per state row; lo: bits 3..7 zero (column),
// 8..17 absolute table row of the next state (row x 256 = its LDS byte address: the table starts at LDS offset 0), 18..19 zero, 20.. as WIDE; hi as WIDE.  The column of a
// byte comes from the def's 256-byte class LUT (class x 8).
constexpr int kCwRowShift = 8;
constexpr uint32_t kCwRowMask = 0x3ffu << kCwRowShift;
constexpr uint32_t kCwClasses = 32u, kCwMaxRows = 1023u;

// HALF table (position-major kernel; all defs together have at most 256 real states and substr ids <= 62): 2-byte
// entries, so that a 256-state x 256-symbol DFA (cfg 5) is LDS-resident in 128 KiB instead of being walked out of L2.
//   bits 0..7   absolute table row of the NEXT state (rows = real states only: no dummy row, no dead row)
//   bits 8..13  substr_id, bit 14 is_start, bit 15 is_end of the next row; 0xff in the high byte = undefined transition
//               (lib.rs:817; not absorbing: the walk goes on from row 0 and the running max of the entries tells)
// LDS byte address of entry (row, c) = (c >> 7) << 16 | row << 8 | (c & 127) << 1: every field is byte-aligned, so the
// next lookup address is ONE v_perm_b32 of the entry and the pre-shifted byte (c << 1).
constexpr uint32_t kHalfDead = 0xff00u;
constexpr uint32_t kHalfUpperBase = 1u << 16;     // LDS byte offset of the columns 128..255
constexpr uint32_t kHalfMaxSid = 62;
HRX_HD uint32_t half_addr(uint32_t row, uint32_t c) { return ((c >> 7) << 16) | (row << 8) | ((c & 127u) << 1); }
HRX_HD uint32_t half_image_bytes(uint32_t rows) { return kHalfUpperBase + rows * 256u; }

// BYTE table (position-major kernel; ONE def of at most 256 table rows — cfg 5's 256-state x 256-symbol DFA): the HALF table's
// 128 KiB leave 32 KiB of LDS, one ring slot per walker and no room for a finisher wave, and in-kernel stamps showed that
// walker spending 46 % of its cycles in the tile-end work (reveal masks, held rows, repairs, masked-row stores).  Here the
// dependent chain reads a 1-byte next-state table — entry (row, c) at LDS byte address row << 8 | c, 64 KiB for 256 rows, one
// v_lshl_or_b32 + one ds_read_u8 per row — and the substring tag of a row, which is a function of the PAIR (state, next)
// (lib.rs:831-840, 861-866, 874-879), comes off the chain from a perfect-hash table: 256 .. 4096 u16 slots {next, substr id << 8,
// is_start << 14, is_end << 15} at slot (state * A + next * B) & (slots - 1), with (A, B) searched at finalize time so that the tagged
// pairs do not collide; the tag counts only if the slot's key is the row's next state (A odd: slot and next determine the state).  One more LDS read per row (a row-displacement table
// with its u16 disp[state] took two: 162 cycles per row against the HALF walk's 113), software-pipelined two rows deep, not on
// the chain.  Rows: the real states, then one absorbing dead row if the DFA is partial (lib.rs:817).
constexpr uint32_t kByteNoDead = 0x100u;
constexpr uint32_t kByteSlots = 4096u, kByteMinSlots = 256u;   // most / fewest pair-tag slots

// PAIR table (position-major kernel hrx_kernel_pp.hip; one def, at most kPairMaxClasses byte-equivalence classes): one
// dependent lookup per TWO input bytes.  Bytes are mapped to classes first (class LUT, value = class * 8); block s holds
// the n_classes^2 entries of state s (real states, then one absorbing dead block), entry (a, b) = the walk from s over a
// byte of class a and then one of class b:
//   lo: bits 0..15  LDS byte address / 8 of the block of the state after BOTH bytes   (next lookup = lo.u16 * 8 + index)
//       bits 16..23 substr_id of the first transition, bits 24..31 substr_id of the second
//   hi: byte 0 state s itself, byte 1 the state between the two bytes, byte 2 is_start | is_end << 1 of the first
//       transition, byte 3 the same of the second
// so that either row's compact record state | substr_id << 16 | flags << 24 is ONE v_perm_b32 of (lo, hi), and the dependent
// chain per two rows is one v_mad_u32_u16 + one ds_read_b64.  An undefined transition enters the dead block (state id
// largest + 1 in the hi bytes).
constexpr uint32_t kPairMaxClasses = 31;          // class * 8 fits the u8 LUT
constexpr uint32_t kPairMaxBytes = 120u * 1024u;  // leaves room for one loader/walker pair with a 2-slot ring
HRX_HD uint32_t pair_index(uint32_t cls8_a, uint32_t cls8_b, uint32_t n_classes) { return cls8_a * n_classes + cls8_b; }  // byte offset inside a block

// Position-major buffers are BLOCKED: strings [k * kPmBlock, (k + 1) * kPmBlock) form block k, and each block is a complete
// position-major array of its own strings ([M/4][D][nb][4] records, [M/8][nb][8] masked, [stride/16][nb][16] input, nb = strings
// in the block), blocks back to back.  A batch of at most kPmBlock strings is one block — the plain layout.  Reason: the
// distance between a string's consecutive quads is nb * 16 bytes; with 2^20 strings in one array that is 16 MiB, every store
// instruction opens another page, and the same kernel ran 14 % slower on 262144 strings than on 4 x 65536.
// 65536 strings = one round of the chip (256 CUs x 4 walkers x 64 lanes).
constexpr uint32_t kPmBlock = 65536u;

// compact witness record (u32): state | substr_id << 16 | start_enable << 24 | end_enable << 25
constexpr uint32_t kRecEndBit = 1u << 25;

// per-string status word (u64), shared with the oracle (oracle/hrx_oracle.c pack_status)
constexpr uint64_t kStatusOk = 0, kStatusInvalidTransition = 1, kStatusFlagOverlap = 2, kStatusBadLength = 3;

HRX_HD uint64_t brev64(uint64_t x) {
#if defined(__clang__)
    return __builtin_bitreverse64(x);  // s_brev_b64 / v_bfrev_b32 on gfx950
#else
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0f0f0f0f0f0f0f0full) | ((x & 0x0f0f0f0f0f0f0f0full) << 4);
    x = ((x >> 8) & 0x00ff00ff00ff00ffull) | ((x & 0x00ff00ff00ff00ffull) << 8);
    x = ((x >> 16) & 0x0000ffff0000ffffull) | ((x & 0x0000ffff0000ffffull) << 16);
    return (x >> 32) | (x << 32);
#endif
}

HRX_HD int ctz64(uint64_t x) { return __builtin_ctzll(x); }        // x != 0
HRX_HD int msb64(uint64_t x) { return 63 - __builtin_clzll(x); }   // index of the highest set bit, x != 0

// "Last event wins" scan towards higher bit positions.
//   out[i] = set[i] ? 1 : rst[i] ? 0 : out[i-1],   out[-1] = cin        (set & rst == 0)
// which is the recurrence  new = select(0, select(1, last, is_set), is_reset)  of
// src/lib.rs:631-642.  Ripple-carry identity: with generate G = set and propagate
// P = ~(set|rst), out[i] is the carry OUT of bit i of (P|G) + G + cin; the carry INTO
// bit i is ((P|G) + G + cin) ^ P.
HRX_HD uint64_t fill_up(uint64_t set, uint64_t rst, uint32_t cin) {
    const uint64_t P = ~(set | rst);
    const uint64_t sum = (P | set) + set + (uint64_t)cin;
    return set | (P & (sum ^ P));
}

// Same towards lower bit positions: out[i] = set[i] ? 1 : rst[i] ? 0 : out[i+1], out[64] = cin
// (src/lib.rs:699-710 runs this from the last row down).
HRX_HD uint64_t fill_down(uint64_t set, uint64_t rst, uint32_t cin) {
    return brev64(fill_up(brev64(set), brev64(rst), cin));
}

// What a lane carries from one tile of its string to the next.
struct MaskCarry {
    uint32_t sm;          // start_mask of the previous row                      (last_start_mask, lib.rs:598,644)
    uint32_t en;          // EN[t0]: an is_end flag lands on the first row of the next tile
    uint32_t pend;        // rows [pend_start, t0) were written assuming end_mask = 1
    uint32_t pend_start;
};

// Bitvectors of one tile, bit p <-> row r = t0 + p  (SURVEY App. A.2/A.3 notation):
//   st  : ST[r]   = sum_d is_start_d[r]  != 0           (0 for r >= n)
//   en1 : EN[r+1] = sum_d is_end_d[r+1]  != 0           (0 unless r < n and r <= M-2; lib.rs:501-519)
//   ch  : SID[r] != SID[r-1],  SID = sum_d substr_id_d, SID[-1] = 0, SID[r>=n] = 0
struct TileBits {
    uint64_t st, en1, ch;
};

struct TileMasks {
    uint64_t mask;        // start_mask & end_mask for rows r < n
    uint32_t fix;         // 1: zero masked rows [fix_start, t0) — an earlier optimistic end_mask = 1 was wrong
    uint32_t fix_start;
    // for walks that start in the middle of a string (chunks, hrx_kernel_spec.hip): what this tile says about rows BEFORE it
    uint32_t dec;         // 0: nothing (no backward event, not exact); 1 / 2: end_mask of row t0 - 1 is 1 / 0 — what decides rows left pending before t0
    uint32_t fwd;         // 1: the tile holds a forward (start_mask) event: the carry out of it does not depend on the carry into it
};

// Reveal mask of one tile of W rows, W <= 64 (src/lib.rs:598-764 on integers); bits >= W of the inputs are 0.
//
// The backward scan (end_mask) of row p depends on rows > p.  The backward event that
// decides position q-1 is made of row q's own quantities (lib.rs:665-698):
//     set   = EN[q] & (SID[q] != SID[q-1]),   reset = !EN[q] & ST[q] & (SID[q] != SID[q-1])
// so all events of positions t0-1 .. t0+W-2 are known inside the tile; what is not
// known is the first event at or after t0+W-1.  `exact` says there is none
// (the string or the region ends inside/before this tile: no events at rows > n, none at row M).
// Otherwise the rows after the tile's last event are emitted optimistically with
// end_mask = 1 (the common case: a substring that ends in the next tile) and the lane
// remembers where they start; the first event of a later tile confirms them or asks for
// them to be zeroed (`fix`).  Each row is fixed at most once.
// word type wide enough for a W-row tile: the walker/storer kernel's 32- and 16-row tiles run the whole algebra in
// 32-bit registers (half the VALU work of the 64-bit form)
template <int W> struct TileWord { typedef uint64_t type; };
template <> struct TileWord<32> { typedef uint32_t type; };
template <> struct TileWord<16> { typedef uint32_t type; };

template <class U> HRX_HD U brev_word(U x);
template <> HRX_HD uint64_t brev_word<uint64_t>(uint64_t x) { return brev64(x); }
template <> HRX_HD uint32_t brev_word<uint32_t>(uint32_t x) {
#if defined(__clang__)
    return __builtin_bitreverse32(x);
#else
    return (uint32_t)(brev64((uint64_t)x) >> 32);
#endif
}
template <class U> HRX_HD U fill_up_word(U set, U rst, uint32_t cin) {
    const U P = (U) ~(set | rst);
    const U sum = (U)((U)(P | set) + set + (U)cin);
    return (U)(set | (P & (sum ^ P)));
}

template <int W = 64>
HRX_HD TileMasks tile_masks(const TileBits &b, MaskCarry &c, uint32_t t0, bool exact, uint64_t valid) {
    typedef typename TileWord<W>::type U;
    constexpr int BITS = (int)sizeof(U) * 8;
    constexpr U kAll = W == BITS ? (U) ~(U)0 : (U)(((U)1 << (W & (BITS - 1))) - (U)1);
    const U st = (U)b.st, en1 = (U)b.en1, ch = (U)b.ch;
    TileMasks out;
    const U en0 = (U)(((U)(en1 << 1) | (U)c.en) & kAll);  // bit p = EN[t0+p]
    c.en = (uint32_t)(en1 >> (W - 1)) & 1u;
    // forward: start_mask                                                     lib.rs:598-645
    const U setF = (U)(st & ch);
    const U rstF = (U)(~st & en0 & ch);
    const U sm = (U)(fill_up_word<U>(setF, rstF, c.sm) & kAll);
    c.sm = (uint32_t)(sm >> (W - 1)) & 1u;
    // backward: end_mask; bit j of setB/rstB is the event of position t0+j-1   lib.rs:663-714
    const U setB = (U)(en0 & ch);
    const U rstB = (U)(~en0 & st & ch);
    const uint32_t cin = exact ? 0u : 1u;
    // fill towards lower positions over W bits: mirror the W-bit field, fill up, mirror back
    const U F = brev_word<U>((U)(fill_up_word<U>((U)(brev_word<U>(setB) >> (BITS - W)), (U)(brev_word<U>(rstB) >> (BITS - W)), cin)
                                 << (BITS - W)));
    const U em = (U)((U)(F >> 1) | (U)((U)cin << (W - 1)));  // end_mask[t0+p] = F[p+1]
    const U mask = (U)(sm & em & (U)valid & kAll);         // lib.rs:740-745
    out.mask = mask;
    const U any = (U)(setB | rstB);
    out.fix = 0;
    out.fix_start = 0;
    out.dec = (any != 0 || exact) ? ((F & 1) ? 1u : 2u) : 0u;
    out.fwd = (U)(setF | rstF) != 0 ? 1u : 0u;
    if (c.pend && (any != 0 || exact)) {
        if (!(F & 1)) { out.fix = 1; out.fix_start = c.pend_start; }
        c.pend = 0;
    }
    if (!exact) {
        const U suffix = any ? (U)((U) ~(U)0 << msb64((uint64_t)any)) : (U) ~(U)0;  // rows whose end_mask came from cin
        const U opt = (U)(mask & suffix);
        if (opt != 0 && !c.pend) { c.pend = 1; c.pend_start = t0 + (uint32_t)ctz64((uint64_t)opt); }
    }
    return out;
}

// No backward event can exist beyond this W-row tile (rows > n carry no flags, row M does not exist).
HRX_HD bool tile_is_exact(uint32_t t0, uint32_t n, uint32_t M, uint32_t W = 64) { return n <= t0 + W - 1u || t0 + W >= M; }

// rows r = t0+p with r < n, as a bitvector
HRX_HD uint64_t rows_below(uint32_t t0, uint32_t n) {
    if (n <= t0) return 0;
    const uint32_t k = n - t0;
    return k >= 64 ? ~0ull : ((1ull << k) - 1);
}

HRX_HD uint64_t status_ok(uint32_t accept_mask) { return (uint64_t)accept_mask << 8; }   // bit 8 + d: def d ends in its accept state (up to 32 defs)
HRX_HD uint64_t status_invalid(uint32_t def, uint32_t pos, uint32_t state, uint32_t ch) {
    return kStatusInvalidTransition | (uint64_t)(def & 0xff) << 8 | (uint64_t)(ch & 0xff) << 16 |
           (uint64_t)(state & 0xffff) << 24 | (uint64_t)(pos & 0xffffff) << 40;
}
HRX_HD uint64_t status_overlap(uint32_t row) { return kStatusFlagOverlap | (uint64_t)(row & 0xffffff) << 40; }

}  // namespace hrx
