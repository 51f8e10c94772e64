// lane_sim.cpp — CPU re-enactment of ONE LANE of the witness kernel.  TEST INFRASTRUCTURE ONLY.
//
// Compiled by g++ from tests/test_lane_sim.py.  It drives the product's own host code — the dense
// fused-table builder (csrc/hrx_defs.cpp) and the per-lane tile algebra incl. the optimistic end-mask
// protocol (csrc/hrx_lane.h: tile_masks, fill_up/fill_down, status packers) — tile by tile exactly as
// the kernel does (csrc/hrx_kernel_sm.hip: walk_tile<D,false> + the per-tile epilogue), so that the
// algorithm can be checked against the oracle without a GPU.  It is not a fallback: nothing in the
// product loads it.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../halo2_regex_amd/csrc/hrx_defs.hpp"
#include "../../halo2_regex_amd/csrc/hrx_lane.h"

using namespace hrx;

extern "C" {

void *sim_new() { return new DefsSet(); }
void sim_free(void *p) { delete (DefsSet *)p; }
int sim_push_allstr(void *p, const char *t, size_t n) {
    RegexDefs rd;
    if (parse_allstr_text(t, n, rd.allstr)) return 1;
    ((DefsSet *)p)->defs.push_back(std::move(rd));
    return 0;
}
int sim_push_substr(void *p, const char *t, size_t n) {
    SubstrRegexDef sd;
    if (parse_substr_text(t, n, sd)) return 1;
    ((DefsSet *)p)->defs.back().substrs.push_back(std::move(sd));
    return 0;
}
int sim_finalize(void *p) {
    std::string err;
    return finalize_defs(*(DefsSet *)p, err);
}

}  // extern "C"

// Same buffers as hrx_witness_batch_host.  fixups (optional) counts optimistic rows that had to be zeroed.
template <int W>
static void sim_batch_w(void *p, const uint8_t *chars, size_t stride, const uint32_t *lens, size_t B, size_t M,
                        uint32_t *records, uint16_t *masked, uint64_t *status, uint64_t *fixups) {
    const DefsSet &s = *(DefsSet *)p;
    const int D = (int)s.defs.size();
    const uint32_t *T = s.table_image.data();
    const uint32_t ntiles = (uint32_t)((M + W - 1) / W);
    uint64_t nfix = 0;
    for (size_t b = 0; b < B; ++b) {
        const uint8_t *cp = chars + b * stride;
        const uint32_t n_raw = lens[b];
        const bool badlen = n_raw > M;
        const uint32_t n = badlen ? (uint32_t)M : n_raw;
        uint32_t e[3], mx[3] = {0, 0, 0};
        for (int d = 0; d < D; ++d) e[d] = s.consts[d].first_entry;
        uint32_t sid_prev = 0, ov_row = 0xffffffffu, dead = 0, accept = 0;
        uint32_t err_pos[3] = {0, 0, 0}, err_state[3] = {0, 0, 0}, err_char[3] = {0, 0, 0};
        MaskCarry mc = {0, 0, 0, 0};
        uint32_t *rec = records + b * M * D;
        uint16_t *msk = masked + b * M;
        for (uint32_t t = 0; t < ntiles; ++t) {
            const uint32_t t0 = t * W;
            const int rem = (int)n - (int)t0, mrem = (int)M - 1 - (int)t0;
            uint32_t trec[W * 3];
            uint8_t tch[W];
            TileBits tb = {0, 0, 0};
            for (int p = 0; p < W; ++p) {
                const uint32_t r = t0 + p;
                const uint8_t c = (r < n) ? cp[r] : 0;  // the kernel zero-fills chunks it does not load
                tch[p] = c;
                const bool live = p < rem;
                uint32_t sid = 0, stn = 0, enn = 0;
                for (int d = 0; d < D; ++d) {
                    const uint32_t state = (e[d] >> kNextShift) - s.consts[d].row_base;
                    uint32_t ent = T[((e[d] & ~kTagMask) | ((uint32_t)c << 2)) / 4];
                    ent = live ? ent : s.consts[d].dummy_entry;
                    uint32_t tag = ent & kTagMask;
                    if (p >= mrem) tag &= ~kTagEnd;
                    trec[p * D + d] = state | (tag << 16);
                    e[d] = ent;
                    if (ent > mx[d]) mx[d] = ent;
                    sid += tag & 0xff; stn += (tag >> 8) & 1; enn += (tag >> 9) & 1;
                }
                if (D > 1) {
                    if (stn > 1 && r < ov_row) ov_row = r;
                    if (enn > 1 && r + 1 < ov_row) ov_row = r + 1;
                }
                tb.st |= (uint64_t)(stn ? 1 : 0) << p;
                tb.en1 |= (uint64_t)(enn ? 1 : 0) << p;
                tb.ch |= (uint64_t)(sid != sid_prev ? 1 : 0) << p;
                sid_prev = sid;
            }
            for (int d = 0; d < D; ++d) {
                if (!((dead >> d) & 1) && mx[d] >= s.consts[d].dead_entry) {
                    const uint32_t dead_state = s.consts[d].n_rows - 1;
                    for (uint32_t p = 0; p < (uint32_t)W; ++p) {
                        const uint32_t s_p = trec[p * D + d] & 0xffff;
                        const uint32_t s_n = p < (uint32_t)W - 1 ? (trec[(p + 1) * D + d] & 0xffff) : ((e[d] >> kNextShift) - s.consts[d].row_base);
                        if (s_n == dead_state && s_p != dead_state) { err_pos[d] = t0 + p; err_state[d] = s_p; err_char[d] = tch[p]; break; }
                    }
                    dead |= 1u << d;
                }
            }
            if (n >= t0 && n < t0 + W) {
                accept = 0;
                for (int d = 0; d < D; ++d) accept |= ((trec[(n - t0) * D + d] & 0xffff) == s.consts[d].accepted_state ? 1u : 0u) << d;
            } else if (n == t0 + W && t + 1 == ntiles) {
                accept = 0;
                for (int d = 0; d < D; ++d) accept |= (((e[d] >> kNextShift) - s.consts[d].row_base) == s.consts[d].accepted_state ? 1u : 0u) << d;
            }
            const TileMasks tm = tile_masks<W>(tb, mc, t0, tile_is_exact(t0, n, (uint32_t)M, W), rows_below(t0, n));
            if (tm.fix) {
                for (uint32_t r = tm.fix_start; r < t0; ++r) { if (msk[r]) ++nfix; msk[r] = 0; }
            }
            for (uint32_t p = 0; p < (uint32_t)W && t0 + p < M; ++p) {
                uint32_t sid = 0;
                for (int d = 0; d < D; ++d) { rec[(size_t)(t0 + p) * D + d] = trec[p * D + d]; sid += (trec[p * D + d] >> 16) & 0xff; }
                msk[t0 + p] = ((tm.mask >> p) & 1) ? (uint16_t)(tch[p] | (sid << 8)) : 0;
            }
        }
        uint64_t sw;
        if (badlen) sw = kStatusBadLength;
        else if (dead) { sw = 0; for (int d = D - 1; d >= 0; --d) if ((dead >> d) & 1) sw = status_invalid(d, err_pos[d], err_state[d], err_char[d]); }
        else if (D > 1 && ov_row != 0xffffffffu) sw = status_overlap(ov_row);
        else sw = status_ok(accept);
        status[b] = sw;
    }
    if (fixups) *fixups = nfix;
}

extern "C" {

void sim_witness_batch(void *p, const uint8_t *chars, size_t stride, const uint32_t *lens, size_t B, size_t M,
                       uint32_t *records, uint16_t *masked, uint64_t *status, uint64_t *fixups) {
    sim_batch_w<64>(p, chars, stride, lens, B, M, records, masked, status, fixups);
}
// tile width of the walker/storer kernel: 32 rows (D = 1) or 16 rows (D = 2)
void sim_witness_batch_w(void *p, int W, const uint8_t *chars, size_t stride, const uint32_t *lens, size_t B, size_t M,
                         uint32_t *records, uint16_t *masked, uint64_t *status, uint64_t *fixups) {
    if (W == 32) sim_batch_w<32>(p, chars, stride, lens, B, M, records, masked, status, fixups);
    else if (W == 16) sim_batch_w<16>(p, chars, stride, lens, B, M, records, masked, status, fixups);
    else sim_batch_w<64>(p, chars, stride, lens, B, M, records, masked, status, fixups);
}

// HALF image (2-byte entries, hrx_lane.h) against the 4-byte fused table, entry by entry.
// Returns the number of (state, byte) entries compared, -1 if the defs have no HALF image, -2 - index on a mismatch.
long sim_check_half_image(void *p) {
    const DefsSet &s = *(DefsSet *)p;
    if (s.half_image.empty()) return -1;
    long n = 0;
    for (size_t d = 0; d < s.defs.size(); ++d) {
        const DefConsts &c = s.consts[d];
        for (uint32_t st = 0; st + 3 <= c.n_rows; ++st)   // real states only
            for (uint32_t ch = 0; ch < 256; ++ch, ++n) {
                const uint32_t e4 = s.table_image[(size_t)(c.row_base + st) * 256 + ch];
                const uint32_t e2 = s.half_image[half_addr(c.half_row_base + st, ch) / 2];
                if ((e4 & ~kTagMask) == c.dead_entry) {
                    if (e2 < kHalfDead) return -2 - n;
                    continue;
                }
                const uint32_t tag = ((e2 >> 8) & 0x3fu) | ((e2 >> 14) << 8);
                if (e2 >= kHalfDead || (e2 & 0xffu) - c.half_row_base != (e4 >> kNextShift) - c.row_base || tag != (e4 & kTagMask)) return -2 - n;
            }
    }
    return n;
}

// BYTE image (1-byte next states + the pair tags in a perfect-hash table, hrx_lane.h) against the 4-byte fused table: every
// (state, byte) — the kernel's lookups restated: next = image[state << 8 | byte], slot = ptab[(state * A + next * B) & (slots - 1)],
// tag = slot.next == next ? slot.tag : 0.  Returns the number of entries compared, -1 without a BYTE image, -2 - index on a mismatch.
long sim_check_byte_image(void *p) {
    const DefsSet &s = *(DefsSet *)p;
    if (s.byte.image.empty()) return -1;
    const DefConsts &c = s.consts[0];
    const ByteTable &b = s.byte;
    if (b.slots < kByteMinSlots || b.slots > kByteSlots || (b.slots & (b.slots - 1)) || b.ptab_off % (b.slots * 4) || b.ptab_off + b.slots * 4 != b.bytes || !(b.mul_a & 1u)) return -2;
    long n = 0;
    for (uint32_t st = 0; st < b.n_rows; ++st)
        for (uint32_t ch = 0; ch < 256; ++ch, ++n) {
            const uint32_t nx = b.image[(size_t)st << 8 | ch];
            uint32_t slot;
            std::memcpy(&slot, &b.image[b.ptab_off + (((st * b.mul_a + nx * b.mul_b) & (b.slots - 1)) << 2)], 4);
            const uint32_t tag = (slot & 0xffu) == nx ? ((slot >> 8) & 0x3fu) | ((slot >> 14) & 3u) << 8 : 0u;   // the tag byte: what the finisher gets
            if ((slot & 0xffu) == nx && (slot >> 16) != tag) return -2 - n;                                       // the record half: what the walker stores
            if ((slot & 0xffu) != nx && slot != 0 && ((slot & 0xffu) >= b.n_rows)) return -2 - n;
            if (st == b.dead) {                                   // the absorbing dead row (partial DFAs only)
                if (nx != b.dead || tag) return -2 - n;
                continue;
            }
            const uint32_t e4 = s.table_image[(size_t)(c.row_base + st) * 256 + ch];
            if ((e4 & ~kTagMask) == c.dead_entry) {
                if (nx != b.dead || b.dead == kByteNoDead) return -2 - n;
                continue;
            }
            if (nx != (e4 >> kNextShift) - c.row_base || tag != (e4 & kTagMask)) return -2 - n;
        }
    return n;
}

// direct access to the scan primitives for property tests
uint64_t sim_fill_up(uint64_t set, uint64_t rst, uint32_t cin) { return fill_up(set, rst, cin); }
uint64_t sim_fill_down(uint64_t set, uint64_t rst, uint32_t cin) { return fill_down(set, rst, cin); }

}  // extern "C"
