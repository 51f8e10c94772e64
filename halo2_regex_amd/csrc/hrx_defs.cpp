// hrx_defs.cpp — text parsers of src/defs.rs and the dense fused-table builder.
#include "hrx_defs.hpp"

#include <algorithm>
#include <map>
#include <cmath>
#include <cstring>

#include "../../include/hrx.h"
#include "hrx_lane.h"

namespace hrx {

// One line -> Vec<u64>, like `line.split_whitespace().map(|s| s.parse::<u64>())` (defs.rs:86-92).
// false where the reference's parse().expect() panics.
static bool split_u64(const char *p, const char *end, std::vector<uint64_t> &out) {
    out.clear();
    auto is_ws = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; };
    while (p < end) {
        while (p < end && is_ws(*p)) ++p;
        if (p >= end) break;
        if (*p == '+') ++p;
        if (p >= end || *p < '0' || *p > '9') return false;
        uint64_t v = 0;
        while (p < end && *p >= '0' && *p <= '9') {
            const uint64_t d = (uint64_t)(*p - '0');
            if (v > (UINT64_MAX - d) / 10) return false;
            v = v * 10 + d;
            ++p;
        }
        if (p < end && !is_ws(*p)) return false;
        out.push_back(v);
    }
    return true;
}

template <class F>
static int for_each_line(const char *text, size_t len, F &&f) {
    const char *p = text, *end = text + len;
    uint64_t idx = 0;
    std::vector<uint64_t> el;
    while (p < end) {  // BufRead::lines(): no extra empty line after a trailing '\n'
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        if (!split_u64(p, le, el)) return -(int)(idx + 1);
        if (!f(idx, el)) return -(int)(idx + 1);
        ++idx;
        p = nl ? nl + 1 : end;
    }
    return 0;
}

// AllstrRegexDef::read_from_reader — src/defs.rs:75-110
int parse_allstr_text(const char *text, size_t len, AllstrRegexDef &out) {
    out = AllstrRegexDef();
    return for_each_line(text, len, [&](uint64_t idx, const std::vector<uint64_t> &el) {
        if (idx <= 2) {
            if (el.empty()) return false;
            if (idx == 0) out.first_state_val = el[0];
            else if (idx == 1) out.accepted_state_val = el[0];
            else out.largest_state_val = el[0];
        } else {
            if (el.size() < 3) return false;
            // insert((elements[2] as u8, elements[0]), (idx, elements[1])): a later duplicate overwrites (defs.rs:100)
            out.state_lookup[{(uint64_t)(uint8_t)el[2], el[0]}] = AllstrRegexDef::Val{idx, el[1]};
        }
        return true;
    });
}

// SubstrRegexDef::read_from_reader — src/defs.rs:209-265
int parse_substr_text(const char *text, size_t len, SubstrRegexDef &out) {
    out = SubstrRegexDef();
    return for_each_line(text, len, [&](uint64_t idx, const std::vector<uint64_t> &el) {
        if (idx <= 2) {
            if (el.empty()) return false;
            if (idx == 0) out.max_length = el[0];
            else if (idx == 1) out.min_position = el[0];
            else out.max_position = el[0];
        } else if (idx == 3) {
            out.start_states = el;
        } else if (idx == 4) {
            out.end_states = el;
        } else {
            if (el.size() < 2) return false;
            out.valid_state_transitions.insert({el[0], el[1]});
        }
        return true;
    });
}

// tag of the transition (cur -> next) of def `rd`: substr id of the FIRST substring whose
// valid_state_transitions holds the pair (lib.rs:831-840, table.rs:110-120), start flag if cur is one of
// that substring's start_states (lib.rs:861-866), end flag if next is one of its end_states (lib.rs:874-879).
static uint32_t pair_tag(const RegexDefs &rd, uint64_t off, uint64_t cur, uint64_t next) {
    for (size_t j = 0; j < rd.substrs.size(); ++j) {
        const SubstrRegexDef &sd = rd.substrs[j];
        if (sd.valid_state_transitions.count({cur, next})) {
            uint32_t tag = (uint32_t)(off + j);
            if (std::find(sd.start_states.begin(), sd.start_states.end(), cur) != sd.start_states.end()) tag |= kTagStart;
            if (std::find(sd.end_states.begin(), sd.end_states.end(), next) != sd.end_states.end()) tag |= kTagEnd;
            return tag;
        }
    }
    return 0;
}

// PAIR image (hrx_lane.h) from the dense 4-byte table of def 0: byte classes = identical columns over the real states.
static void build_pair_table(DefsSet &s) {
    s.pair = PairTable();
    if (s.defs.size() != 1) return;
    const DefConsts &c = s.consts[0];
    const uint32_t L = (uint32_t)s.defs[0].allstr.largest_state_val;
    if (L + 1 > 254) return;
    const uint32_t *T = s.table_image.data() + (size_t)c.row_base * 256;
    // classes in order of their first byte
    std::vector<int> cls(256, -1);
    std::vector<int> rep;
    for (int ch = 0; ch < 256; ++ch) {
        for (size_t k = 0; k < rep.size() && cls[ch] < 0; ++k) {
            bool same = true;
            for (uint32_t st = 0; st <= L && same; ++st) same = T[st * 256 + ch] == T[st * 256 + rep[k]];
            if (same) cls[ch] = (int)k;
        }
        if (cls[ch] < 0) { cls[ch] = (int)rep.size(); rep.push_back(ch); }
    }
    const uint32_t C = (uint32_t)rep.size();
    if (C > kPairMaxClasses) return;
    PairTable p;
    p.n_classes = C;
    p.n_blocks = L + 2;
    p.blk_bytes = C * C * 8;
    p.lut_off = p.n_blocks * p.blk_bytes;
    p.bytes = (p.lut_off + 256 + 15) & ~15u;
    if (p.bytes > kPairMaxBytes || p.lut_off / 8 > 0xffffu) return;
    p.image.assign(p.bytes, 0);
    const uint32_t dead = L + 1;
    auto step = [&](uint32_t st, int ch, uint32_t &next, uint32_t &tag) {   // delta + tag of the fused 4-byte entry
        if (st == dead) { next = dead; tag = 0; return; }
        const uint32_t e = T[st * 256 + ch];
        if (e >= c.dead_entry) { next = dead; tag = 0; return; }
        next = (e >> kNextShift) - c.row_base;
        tag = e & kTagMask;
    };
    for (uint32_t st = 0; st <= dead; ++st)
        for (uint32_t a = 0; a < C; ++a)
            for (uint32_t b = 0; b < C; ++b) {
                uint32_t mid, nxt, t0, t1;
                step(st, rep[a], mid, t0);
                step(mid, rep[b], nxt, t1);
                const uint64_t lo = (uint64_t)(nxt * p.blk_bytes / 8) | (uint64_t)(t0 & 0xffu) << 16 | (uint64_t)(t1 & 0xffu) << 24;
                const uint64_t hi = (uint64_t)st | (uint64_t)mid << 8 | (uint64_t)((t0 >> 8) & 3u) << 16 | (uint64_t)((t1 >> 8) & 3u) << 24;
                const uint64_t e = lo | hi << 32;
                std::memcpy(&p.image[(size_t)st * p.blk_bytes + (a * C + b) * 8], &e, 8);
            }
    for (int ch = 0; ch < 256; ++ch) p.image[p.lut_off + ch] = (uint8_t)(cls[ch] * 8);
    s.pair = std::move(p);
}

// BYTE image (hrx_lane.h) of def 0 from its dense 4-byte table and its (state, next) -> tag matrix.
// The tags go into a table of 256 .. 4096 four-byte slots addressed by an ARITHMETIC perfect hash of the pair: slot = (state * A + next * B) & (slots - 1)
// with (A, B) searched so that no two tagged pairs share a slot; a lookup reads the slot of ANY pair and takes the tag only if
// the slot's key is the pair's next state (A is odd and slots >= 256: slot and next determine the state).  Substr ids of 64 and more (a slot has 6 bits for the id), or no (A, B) within the search budget (hundreds of
// tagged pairs): no BYTE image — the HALF table or the global table serves the def.
static void build_byte_table(DefsSet &s) {
    s.byte = ByteTable();
    if (s.defs.size() != 1) return;
    const DefConsts &c = s.consts[0];
    const uint32_t L = (uint32_t)s.defs[0].allstr.largest_state_val, S = L + 1;
    const uint32_t *T = s.table_image.data() + (size_t)c.row_base * 256;
    bool total = true;
    for (uint32_t st = 0; st < S && total; ++st)
        for (int ch = 0; ch < 256 && total; ++ch) total = T[st * 256 + ch] < c.dead_entry;
    const uint32_t rows = S + (total ? 0u : 1u);
    if (rows > 256) return;
    ByteTable b;
    b.n_rows = rows;
    b.dead = total ? kByteNoDead : S;
    const std::vector<uint16_t> &pt = s.pair_tags[0];   // [(L+1)^2]
    std::vector<uint32_t> keys;                          // state << 8 | next of every tagged pair
    bool narrow_ids = true;                              // a slot has 6 bits for the substr id (like the HALF entry)
    for (uint32_t cur = 0; cur < S; ++cur)
        for (uint32_t nx = 0; nx < S; ++nx)
            if (pt[(size_t)cur * S + nx]) { keys.push_back(cur << 8 | nx); narrow_ids = narrow_ids && (pt[(size_t)cur * S + nx] & 0xffu) < 64u; }
    if (!narrow_ids) return;
    // the smallest table (256 .. kByteSlots slots) that is at most a quarter full and has a collision-free (A, B) within the budget
    std::vector<uint32_t> seen(kByteSlots, 0xffffffffu);
    uint32_t A = 0, B = 0, stamp = 0, slots = 0;
    bool found = false;
    for (uint32_t ns = kByteMinSlots; ns <= kByteSlots && !found; ns <<= 1) {
        if (ns < kByteSlots && keys.size() * 4 > ns) continue;
        if (keys.empty()) { A = 1; B = 1; slots = ns; found = true; break; }
        // a random (A, B) is collision-free with probability ~exp(-k (k - 1) / (2 ns)): sizes that leave the budget less than a few expected hits are skipped
        const uint32_t budget = 200000u;
        if (ns < kByteSlots && std::exp(-(double)keys.size() * (double)(keys.size() - 1) / (2.0 * ns)) * budget < 4.0) continue;
        // deterministic search: odd A (so that slot and next determine the state: a slot's key is the next state alone), odd B, in a fixed pseudo-random order
        uint64_t x = 0x9e3779b97f4a7c15ull;
        for (uint32_t tries = 0; !found && tries < budget; ++tries, ++stamp) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            const uint32_t a = ((uint32_t)(x >> 20) & 0xfffu) | 1u, bb = ((uint32_t)(x >> 40) & 0xfffu) | 1u;
            bool ok = true;
            for (uint32_t k : keys) {
                const uint32_t slot = ((k >> 8) * a + (k & 0xffu) * bb) & (ns - 1);
                if (seen[slot] == stamp) { ok = false; break; }
                seen[slot] = stamp;
            }
            if (ok) { A = a; B = bb; slots = ns; found = true; }
        }
    }
    if (!found) return;
    b.mul_a = A; b.mul_b = B; b.slots = slots;
    b.ptab_off = (rows * 256 + slots * 4 - 1) & ~(slots * 4 - 1);   // aligned to its size: slot address = (hash & (slots - 1) * 4) | ptab_off, one v_and_or_b32
    b.bytes = b.ptab_off + slots * 4;
    b.ptab16_off = (rows * 256 + slots * 2 - 1) & ~(slots * 2 - 1);
    b.bytes16 = b.ptab16_off + slots * 2;
    b.image.assign((size_t)b.bytes + slots * 2, 0);
    for (uint32_t st = 0; st < rows; ++st)
        for (int ch = 0; ch < 256; ++ch) {
            uint32_t nx = b.dead;
            if (st < S) {
                const uint32_t e = T[st * 256 + ch];
                if (e < c.dead_entry) nx = (e >> kNextShift) - c.row_base;
            }
            b.image[(size_t)st * 256 + ch] = (uint8_t)nx;   // (total: every entry is a real state; partial: dead = S <= 255)
        }
    // slot (hrx_lane.h): next | tag byte << 8 | record half << 16.  An empty slot is 0: whatever pair reads it gets tag 0 — what an untagged pair has.
    std::vector<uint32_t> slot(slots, 0);
    for (uint32_t k : keys) {
        const uint32_t t = pt[(size_t)(k >> 8) * S + (k & 0xffu)];     // 10-bit tag: id | is_start << 8 | is_end << 9
        slot[((k >> 8) * A + (k & 0xffu) * B) & (slots - 1)] = (k & 0xffu) | (t & 0x3fu) << 8 | (t >> 8) << 14 | t << 16;
    }
    std::memcpy(&b.image[b.ptab_off], slot.data(), (size_t)slots * 4);
    for (uint32_t i = 0; i < slots; ++i) { const uint16_t lo = (uint16_t)slot[i]; std::memcpy(&b.image[(size_t)b.bytes + 2 * i], &lo, 2); }
    s.byte = std::move(b);
}

int finalize_defs(DefsSet &s, std::string &err) {
    if (s.finalized) return HRX_OK;
    if (s.defs.empty()) { err = "no RegexDefs pushed"; return HRX_ERR_STATE; }
    if (s.defs.size() > kMaxDefs) { err = "at most " + std::to_string(kMaxDefs) + " RegexDefs per config (the status word's accept mask)"; return HRX_ERR_BOUNDS; }
    const bool passes = s.defs.size() > kMaxDefsPerPass;   // walked in groups; the kernel-side images below belong to the groups then
    s.consts.clear();
    s.pair_tags.clear();
    s.endpoint_member.clear();
    size_t total_rows = 0, half_rows = 0;
    uint64_t off = s.sid_base;  // substr_id_offset, lib.rs:780,827,854
    uint64_t max_sid_sum = 0;  // largest value Σ_d substr_id_d can take (masked_substr_id is a u8)
    for (size_t d = 0; d < s.defs.size(); ++d) {
        const RegexDefs &rd = s.defs[d];
        const uint64_t L = rd.allstr.largest_state_val;
        if (L > 60000) { err = "largest_state_val too large for a u16 state record"; return HRX_ERR_BOUNDS; }
        if (rd.allstr.first_state_val > L || rd.allstr.accepted_state_val > L) {
            err = "first/accepted state exceeds largest_state_val"; return HRX_ERR_BOUNDS;
        }
        for (const auto &kv : rd.allstr.state_lookup) {
            if (kv.first.second > L || kv.second.next > L) { err = "a transition references a state above largest_state_val"; return HRX_ERR_BOUNDS; }
        }
        DefConsts c{};
        c.n_rows = (uint32_t)L + 3;
        c.row_base = (uint32_t)total_rows;
        c.first_entry = (uint32_t)(total_rows + rd.allstr.first_state_val) << kNextShift;
        c.dummy_entry = (uint32_t)(total_rows + L + 1) << kNextShift;
        c.dead_entry = (uint32_t)(total_rows + L + 2) << kNextShift;
        c.accepted_state = (uint32_t)rd.allstr.accepted_state_val;
        c.substr_id_offset = (uint32_t)off;
        c.half_row_base = (uint32_t)half_rows;
        c.first_state = (uint32_t)rd.allstr.first_state_val;
        c.dummy_state = (uint32_t)L + 1;
        half_rows += L + 1;
        s.consts.push_back(c);
        total_rows += c.n_rows;
        if (!rd.substrs.empty()) max_sid_sum += off + rd.substrs.size() - 1;
        off += rd.substrs.size();
    }
    if (max_sid_sum > 255) { err = "substring ids do not fit the u8 masked_substr_id"; return HRX_ERR_BOUNDS; }
    if (total_rows * 1024 > kMaxTableBytes) {
        err = "fused (state,char) tables need " + std::to_string(total_rows) + " KiB; limit is " +
              std::to_string(kMaxTableBytes / 1024) + " KiB";
        return HRX_ERR_BOUNDS;
    }
    s.table_image.assign(total_rows * 256, 0);
    for (size_t d = 0; d < s.defs.size(); ++d) {
        const RegexDefs &rd = s.defs[d];
        const DefConsts &c = s.consts[d];
        const uint64_t L = rd.allstr.largest_state_val;
        uint32_t *T = s.table_image.data() + (size_t)c.row_base * 256;
        // undefined (state,char) -> dead (the reference panics, lib.rs:817); dummy and dead rows absorb.
        for (uint64_t st = 0; st <= L; ++st)
            for (int ch = 0; ch < 256; ++ch) T[st * 256 + ch] = c.dead_entry;
        for (int ch = 0; ch < 256; ++ch) {
            T[(L + 1) * 256 + ch] = c.dummy_entry;
            T[(L + 2) * 256 + ch] = c.dead_entry;
        }
        std::vector<uint16_t> pt((L + 1) * (L + 1), 0);
        for (const auto &kv : rd.allstr.state_lookup) {
            const uint64_t ch = kv.first.first, cur = kv.first.second, next = kv.second.next;
            const uint32_t tag = pair_tag(rd, c.substr_id_offset, cur, next);
            T[cur * 256 + ch] = ((uint32_t)(c.row_base + next) << kNextShift) | tag;
        }
        for (uint64_t cur = 0; cur <= L; ++cur)
            for (uint64_t next = 0; next <= L; ++next) pt[cur * (L + 1) + next] = (uint16_t)pair_tag(rd, c.substr_id_offset, cur, next);
        s.pair_tags.push_back(std::move(pt));
        std::vector<uint8_t> mem(rd.substrs.size() * (L + 1) + 1, 0);
        for (size_t j = 0; j < rd.substrs.size(); ++j) {
            for (uint64_t st : rd.substrs[j].start_states) if (st <= L) mem[j * (L + 1) + st] |= 1;
            for (uint64_t en : rd.substrs[j].end_states) if (en <= L) mem[j * (L + 1) + en] |= 2;
        }
        s.endpoint_member.push_back(std::move(mem));
    }
    // WIDE image for the position-major kernel
    s.wide_image.clear();
    bool ascii = total_rows <= 255 && !passes;
    for (const RegexDefs &rd : s.defs)
        for (const auto &kv : rd.allstr.state_lookup) if (kv.first.first >= 128) ascii = false;
    if (ascii) {
        s.wide_image.assign(total_rows * 128, 0);
        for (size_t d = 0; d < s.defs.size(); ++d) {
            const RegexDefs &rd = s.defs[d];
            const DefConsts &c = s.consts[d];
            const uint64_t L = rd.allstr.largest_state_val;
            uint64_t *W = s.wide_image.data() + (size_t)c.row_base * 128;
            const uint64_t dead_lo = (uint64_t)(c.row_base + L + 2) << kWideRowShift, dummy_lo = (uint64_t)(c.row_base + L + 1) << kWideRowShift;
            for (uint64_t st = 0; st <= L + 2; ++st)
                for (int ch = 0; ch < 128; ++ch)
                    W[st * 128 + ch] = st == L + 1 ? (dummy_lo | (L + 1) << 32) : (dead_lo | std::min<uint64_t>(st, L + 1) << 32);
            for (const auto &kv : rd.allstr.state_lookup) {
                const uint64_t ch = kv.first.first, cur = kv.first.second, next = kv.second.next;
                const uint64_t tag = pair_tag(rd, c.substr_id_offset, cur, next);
                const uint64_t sid = tag & 0xff, is_start = (tag >> 8) & 1, is_end = (tag >> 9) & 1;
                const uint64_t lo = (uint64_t)(c.row_base + next) << kWideRowShift | sid << kWideSidShift | is_start << kWideStartShift |
                                    is_end << kWideEndShift;
                W[cur * 128 + ch] = lo | (cur | tag << 16) << 32;
            }
        }
    }
    // HALF image for the position-major kernel: 2-byte entries, real states only
    s.half_image.clear();
    if (half_rows <= 256 && off - 1 <= kHalfMaxSid && !passes) {
        s.half_image.assign(half_image_bytes((uint32_t)half_rows) / 2, (uint16_t)kHalfDead);
        for (size_t d = 0; d < s.defs.size(); ++d) {
            const RegexDefs &rd = s.defs[d];
            const DefConsts &c = s.consts[d];
            for (const auto &kv : rd.allstr.state_lookup) {
                const uint32_t ch = (uint32_t)kv.first.first, cur = (uint32_t)kv.first.second, next = (uint32_t)kv.second.next;
                const uint32_t tag = pair_tag(rd, c.substr_id_offset, cur, next);
                const uint32_t e = (c.half_row_base + next) | (tag & 0x3fu) << 8 | ((tag >> 8) & 3u) << 14;
                s.half_image[half_addr(c.half_row_base + cur, ch) / 2] = (uint16_t)e;
            }
        }
    }
    build_pair_table(s);
    if (!passes) build_byte_table(s);
    // CLASS-WIDE image of a whole config of 4 .. kMaxDefsPerLaunch defs (hrx_defs.hpp)
    s.cw_image.clear(); s.cw_consts.clear(); s.cw_lut_off = 0;
    if ((passes || s.cw_group) && s.defs.size() <= 8 && total_rows <= kCwMaxRows) {
        bool ok = true;
        std::vector<uint8_t> luts(s.defs.size() * 256, 0);
        std::vector<uint64_t> tab((size_t)total_rows * kCwClasses, 0);
        for (size_t d = 0; d < s.defs.size() && ok; ++d) {
            const RegexDefs &rd = s.defs[d];
            const DefConsts &c = s.consts[d];
            const uint64_t L = rd.allstr.largest_state_val;
            const uint32_t *T = s.table_image.data() + (size_t)c.row_base * 256;
            // bytes with the same column over the real states are one class (the dummy and dead rows do not tell bytes apart)
            std::map<std::vector<uint32_t>, uint32_t> cls_of;
            std::vector<int> rep;                  // a representative byte per class
            for (int ch = 0; ch < 256 && ok; ++ch) {
                std::vector<uint32_t> col(L + 1);
                for (uint64_t st = 0; st <= L; ++st) col[st] = T[st * 256 + ch];
                auto it = cls_of.find(col);
                if (it == cls_of.end()) {
                    if (rep.size() >= kCwClasses) { ok = false; break; }
                    it = cls_of.emplace(std::move(col), (uint32_t)rep.size()).first;
                    rep.push_back(ch);
                }
                luts[d * 256 + ch] = (uint8_t)(it->second * 8u);
            }
            if (!ok) break;
            uint64_t *W = tab.data() + (size_t)c.row_base * kCwClasses;
            const uint64_t dead_lo = (uint64_t)(c.row_base + L + 2) << kCwRowShift, dummy_lo = (uint64_t)(c.row_base + L + 1) << kCwRowShift;
            for (uint64_t st = 0; st <= L + 2; ++st)
                for (uint32_t k = 0; k < kCwClasses; ++k) {
                    uint64_t w = st == L + 1 ? (dummy_lo | (L + 1) << 32) : (dead_lo | std::min<uint64_t>(st, L + 1) << 32);
                    if (st <= L && k < rep.size()) {
                        const uint32_t e = T[st * 256 + rep[k]];
                        if (e < c.dead_entry) {     // a defined transition: as the WIDE entry, the row field at bit kCwRowShift
                            const uint64_t next = (e >> kNextShift) - c.row_base, tag = e & kTagMask;
                            const uint64_t sid = tag & 0xff, is_start = (tag >> 8) & 1, is_end = (tag >> 9) & 1;
                            w = ((uint64_t)(c.row_base + next) << kCwRowShift | sid << kWideSidShift | is_start << kWideStartShift | is_end << kWideEndShift) | (st | tag << 16) << 32;
                        }
                    }
                    W[st * kCwClasses + k] = w;
                }
            DefConsts cc = c;
            cc.first_entry = (uint32_t)(c.row_base + rd.allstr.first_state_val) << kCwRowShift;
            cc.dummy_entry = (uint32_t)(c.row_base + L + 1) << kCwRowShift;
            cc.dead_entry = (uint32_t)(c.row_base + L + 2) << kCwRowShift;
            s.cw_consts.push_back(cc);
        }
        if (ok) {
            s.cw_lut_off = (uint32_t)(total_rows * 256);
            s.cw_image.resize((size_t)s.cw_lut_off + luts.size());
            std::memcpy(s.cw_image.data(), tab.data(), (size_t)s.cw_lut_off);
            std::memcpy(s.cw_image.data() + s.cw_lut_off, luts.data(), luts.size());
        } else {
            s.cw_consts.clear();
        }
    }
    // more defs than one launch walks: consecutive groups, each finalized as a DefsSet of its own
    s.groups.clear();
    s.group_first.clear();
    s.cw_groups.clear();
    s.cw_group_first.clear();
    if (passes && !s.cw_group && s.defs.size() > 8) {
        // CW groups: ceil(D / 8) groups of nearly equal size (D = 13: 7 + 6; 16: 8 + 8; 9: 5 + 4), every one of at least four defs (the def-parallel CW kernel's range)
        const size_t ng = (s.defs.size() + 7) / 8, base = s.defs.size() / ng, extra = s.defs.size() % ng;
        bool ok = base >= 4;
        size_t d = 0;
        for (size_t g = 0; g < ng && ok; ++g) {
            const size_t cnt = base + (g < extra ? 1 : 0);
            DefsSet grp;
            grp.defs.assign(s.defs.begin() + (long)d, s.defs.begin() + (long)(d + cnt));
            grp.sid_base = s.consts[d].substr_id_offset;
            grp.cw_group = true;
            std::string e2;
            if (finalize_defs(grp, e2) != HRX_OK || grp.cw_image.empty()) { ok = false; break; }
            s.cw_group_first.push_back((uint32_t)d);
            s.cw_groups.push_back(std::move(grp));
            d += cnt;
        }
        if (!ok) { s.cw_groups.clear(); s.cw_group_first.clear(); }
    }
    if (passes && !s.cw_group) {
        // a group takes up to kMaxDefsPerPass defs while their fused 4-byte tables (1 KiB per row) still fit LDS next to the
        // input rings (96 KiB); a def that is larger than that on its own walks alone
        size_t d = 0;
        while (d < s.defs.size()) {
            size_t cnt = 1, rows = s.consts[d].n_rows;
            while (cnt < kMaxDefsPerPass && d + cnt < s.defs.size() && rows + s.consts[d + cnt].n_rows <= 96) { rows += s.consts[d + cnt].n_rows; ++cnt; }
            DefsSet g;
            g.defs.assign(s.defs.begin() + (long)d, s.defs.begin() + (long)(d + cnt));
            g.sid_base = s.consts[d].substr_id_offset;
            const int rc = finalize_defs(g, err);
            if (rc != HRX_OK) return rc;
            s.group_first.push_back((uint32_t)d);
            s.groups.push_back(std::move(g));
            d += cnt;
        }
    }
    s.finalized = true;
    return HRX_OK;
}

// RegexTableConfig::load, transition table — src/table.rs:68-125
size_t table_transition_rows(const DefsSet &s, size_t d, uint64_t *rows, size_t cap_rows) {
    const RegexDefs &rd = s.defs[d];
    const uint64_t dummy = rd.allstr.largest_state_val + 1;  // table.rs:67
    const uint64_t off = s.consts[d].substr_id_offset;
    const size_t n = 1 + rd.allstr.state_lookup.size();
    if (!rows) return n;
    struct Row { uint64_t line, ch, cur, next; };
    std::vector<Row> v;
    v.reserve(rd.allstr.state_lookup.size());
    for (const auto &kv : rd.allstr.state_lookup) v.push_back({kv.second.line_idx, kv.first.first, kv.first.second, kv.second.next});
    std::sort(v.begin(), v.end(), [](const Row &a, const Row &b) { return a.line < b.line; });  // table.rs:108
    size_t k = 0;
    auto put = [&](uint64_t a, uint64_t b, uint64_t c, uint64_t e) {
        if (k < cap_rows) { rows[4 * k] = a; rows[4 * k + 1] = b; rows[4 * k + 2] = c; rows[4 * k + 3] = e; }
        ++k;
    };
    put(0, dummy, dummy, 0);  // table.rs:101
    for (const Row &r : v) put(r.ch, r.cur, r.next, pair_tag(rd, off, r.cur, r.next) & 0xff);
    return n;
}

// RegexTableConfig::load, endpoint table — src/table.rs:126-196
size_t table_endpoint_rows(const DefsSet &s, size_t d, uint64_t *rows, size_t cap_rows) {
    const RegexDefs &rd = s.defs[d];
    const uint64_t dummy = rd.allstr.largest_state_val + 1;
    const uint64_t off = s.consts[d].substr_id_offset;
    size_t k = 0;
    auto put = [&](uint64_t a, uint64_t b, uint64_t c) {
        if (rows && k < cap_rows) { rows[3 * k] = a; rows[3 * k + 1] = b; rows[3 * k + 2] = c; }
        ++k;
    };
    put(0, dummy, dummy);
    for (size_t j = 0; j < rd.substrs.size(); ++j) {
        for (uint64_t st : rd.substrs[j].start_states) put(off + j, st, dummy);
        for (uint64_t en : rd.substrs[j].end_states) put(off + j, dummy, en);
    }
    return k;
}

}  // namespace hrx
