// hrx_defs.hpp — host data model (src/defs.rs) and the dense fused-table builder.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace hrx {

struct PairHash {
    size_t operator()(const std::pair<uint64_t, uint64_t> &p) const noexcept {
        uint64_t x = p.first * 0x9e3779b97f4a7c15ull ^ (p.second + 0x7f4a7c15ull + (p.first << 6));
        x ^= x >> 31;
        return (size_t)(x * 0xff51afd7ed558ccdull);
    }
};

// SubstrRegexDef — src/defs.rs:115-132
struct SubstrRegexDef {
    uint64_t max_length = 0, min_position = 0, max_position = 0;  // unused by the chip (defs.rs:118-125)
    std::unordered_set<std::pair<uint64_t, uint64_t>, PairHash> valid_state_transitions;
    std::vector<uint64_t> start_states, end_states;
};

// AllstrRegexDef — src/defs.rs:26-36.  state_lookup keyed (char, state) -> (line idx, next).
struct AllstrRegexDef {
    struct Val {
        uint64_t line_idx, next;
    };
    std::unordered_map<std::pair<uint64_t, uint64_t>, Val, PairHash> state_lookup;  // key = (char as u8, cur state)
    uint64_t first_state_val = 0, accepted_state_val = 0, largest_state_val = 0;
};

// RegexDefs — src/defs.rs:17-22
struct RegexDefs {
    AllstrRegexDef allstr;
    std::vector<SubstrRegexDef> substrs;
};

// Per-def constants the kernel needs.  The tables of all defs are stacked into one LDS image of
// 256-entry rows; entries hold ABSOLUTE row numbers (row_base + state) << 10, so the next lookup
// address is (entry & ~0x3ff) | 4*byte for every def (hrx_lane.h).
struct DefConsts {
    uint32_t row_base;     // first row of this def's table inside the image
    uint32_t n_rows;       // largest + 3: real states, dummy (largest+1), dead (largest+2)
    uint32_t first_entry;  // (row_base + first_state_val) << 10
    uint32_t dummy_entry;  // (row_base + largest+1) << 10 : padding rows (lib.rs:413)
    uint32_t dead_entry;   // (row_base + largest+2) << 10 : absorbing sink of undefined transitions (lib.rs:817)
    uint32_t accepted_state;
    uint32_t substr_id_offset;
    uint32_t half_row_base;  // first row of this def inside the HALF image (real states only: largest+1 rows per def)
    uint32_t first_state, dummy_state;  // first_state_val, largest+1
};

// PAIR table (hrx_lane.h; position-major kernel hrx_kernel_pp.hip, one def): two input bytes per dependent lookup.
// Bytes are first mapped to their equivalence class (bytes whose table columns are identical over all real states; all
// bytes no state has a transition for form one class).  Block b = state b (real states 0..L, then the absorbing dead
// block L+1), n_classes^2 8-byte entries each, entry (a, b) at block + (a * n_classes + b) * 8.
struct PairTable {
    uint32_t n_classes = 0;     // <= kPairMaxClasses
    uint32_t n_blocks = 0;      // largest + 2
    uint32_t blk_bytes = 0;     // n_classes^2 * 8
    uint32_t lut_off = 0;       // LDS byte offset of the 256-byte class LUT (value = class * 8), right behind the blocks
    uint32_t bytes = 0;         // size of the LDS image (blocks + LUT), a multiple of 16
    std::vector<uint8_t> image; // the exact LDS image
};

// BYTE table (hrx_lane.h; position-major kernel, one def of at most 256 table rows): a 1-byte next-state table on the walk's
// dependent chain — 64 KiB for a 256-state x 256-symbol DFA where the HALF table takes 128 — and the substring tag of a
// transition, which is a function of the PAIR (state, next), in a perfect-hash table off the chain.
struct ByteTable {
    uint32_t n_rows = 0;      // real states (+ one absorbing dead row if the DFA is partial)
    uint32_t dead = 0x100;    // row number of the dead row; 0x100: the DFA is total, there is none
    uint32_t slots = 0;       // pair-tag slots: a power of two, kByteMinSlots .. kByteSlots
    uint32_t ptab_off = 0;    // LDS byte offset of u32 ptab[slots] (aligned to its size): next | substr id << 8 | is_start << 14 | is_end << 15 | (substr id | is_start << 8 | is_end << 9) << 16 (0: empty)
    uint32_t mul_a = 0, mul_b = 0;   // slot of pair (state, next) = (state * mul_a + next * mul_b) & (slots - 1), mul_a odd: collision-free over the tagged pairs
    uint32_t bytes = 0;       // size of the LDS image, a multiple of 16
    // the walker/storer kernel (string-major outputs, hrx_kernel_sm.hip) stages the same next-state bytes and, instead of the 4-byte
    // slots, their low halves only (next | substr id << 8 | is_start << 14 | is_end << 15): the table is 8 KiB smaller at 4096 slots,
    // which is a fourth walker/storer pair.  The 2-byte slots follow the LDS image in `image` (at image[bytes ..]).
    uint32_t ptab16_off = 0;  // LDS byte offset of u16 ptab16[slots] in THAT kernel's image (aligned to its size)
    uint32_t bytes16 = 0;     // size of that LDS image
    std::vector<uint8_t> image;   // [0, bytes): the position-major kernel's LDS image; [bytes, bytes + slots * 2): ptab16
};

constexpr size_t kMaxDefs = 32;          // RegexDefs per config: the status word's accept mask (bits 8..39, include/hrx.h)
constexpr size_t kMaxDefsPerPass = 3;    // defs one kernel launch walks side by side (the kernels are instantiated for D = 1..3)

struct DefsSet {
    std::vector<RegexDefs> defs;
    bool finalized = false;
    // Vec<RegexDefs> of any length (src/lib.rs:112): more than kMaxDefsPerPass defs are walked in PASSES — consecutive defs
    // in groups of at most kMaxDefsPerPass, each group a complete DefsSet of its own (own table images, substr ids
    // continuing the config's numbering: sid_base = substr_id_offset of its first def), combined per row afterwards
    // (hrx_kernel_mp.hip).  Empty for configs of up to kMaxDefsPerPass defs.
    std::vector<DefsSet> groups;
    std::vector<uint32_t> group_first;   // index of each group's first def
    uint64_t sid_base = 1;               // substr_id_offset of def 0: 1 for a config (lib.rs:780), the running offset for a group
    // CLASS-WIDE image (round 5; configs of kMaxDefsPerPass + 1 .. kMaxDefsPerLaunch defs, every def with at most kCwClasses byte classes): the WIDE entry format over byte CLASSES
    // instead of bytes — 8-byte entries, kCwClasses columns, 256 B per state row, the chain word's row field at bit kCwRowShift (10 bits: all the config's rows, numbered like
    // the narrow table's) — [rows x 256 B | one 256-byte class LUT per def (value = class x 8)].  A whole five-def config is a few dozen KiB of LDS, where its 1-KiB-per-row tables are ~120:
    // the def-parallel kernel walks ALL the defs of a config of six or seven in one launch (hrx_kernel_pmd.hip, CW) instead of passes over groups of three.  Empty: not built.
    // More than kMaxDefsPerLaunch defs: besides the groups of three (`groups`, every layout) the config is cut into CW GROUPS of 4 .. 8 consecutive defs, each with a CLASS-WIDE image of its own — the
    // passes of position-major launches then walk eight defs at a time with the def-parallel kernel (D = 16: two passes instead of six).  Empty: a def has more than 32 byte classes, or <= 8 defs.
    std::vector<DefsSet> cw_groups;
    std::vector<uint32_t> cw_group_first;
    bool cw_group = false;                 // this set IS such a group: finalize builds its CLASS-WIDE image and no groups of its own
    std::vector<uint8_t> cw_image;
    uint32_t cw_lut_off = 0;               // LDS byte offset of def 0's LUT (= rows x 256); def d's at + 256 d
    std::vector<DefConsts> cw_consts;      // the consts the kernel sees: entries in the CW encoding (row << kCwRowShift)
    // dense image: for def d, n_rows x 256 u32 entries at table_base (see hrx_lane.h for the entry format)
    std::vector<uint32_t> table_image;
    // WIDE image (hrx_lane.h): n_rows x 128 u64 entries per def, same row numbering; empty unless every transition
    // symbol is < 128 and the rows fit an 8-bit row number
    std::vector<uint64_t> wide_image;
    // HALF image (hrx_lane.h): the exact LDS image, half_image_bytes(total real states) bytes; empty unless all defs
    // together have <= 256 real states and every substr id is <= kHalfMaxSid
    std::vector<uint16_t> half_image;
    // PAIR image (hrx_lane.h): empty unless there is exactly one def with <= kPairMaxClasses byte classes, <= 254 real
    // states and a table of at most kPairMaxBytes
    PairTable pair;
    // BYTE image (hrx_lane.h): empty unless there is exactly one def whose real states (+ a dead row if it is partial) fit 256
    // rows and whose tagged (state, next) pairs fit the displacement table
    ByteTable byte;
    std::vector<DefConsts> consts;
    // (cur,next) -> {sid, is_start(cur), is_end(next)} per def, for the states-in entry points (lib.rs:825-888)
    std::vector<std::vector<uint16_t>> pair_tags;  // [(largest+1)^2], entry = tag bits as in the fused table
    // membership bytes per def: [n_substrs][largest+1]; bit0: state in start_states, bit1: state in end_states
    std::vector<std::vector<uint8_t>> endpoint_member;
};

// Parsers return 0 or -(line_idx+1) where the reference would panic.
int parse_allstr_text(const char *text, size_t len, AllstrRegexDef &out);
int parse_substr_text(const char *text, size_t len, SubstrRegexDef &out);
// Returns HRX_* status; message in err.
int finalize_defs(DefsSet &s, std::string &err);

size_t table_transition_rows(const DefsSet &s, size_t d, uint64_t *rows4, size_t cap_rows);
size_t table_endpoint_rows(const DefsSet &s, size_t d, uint64_t *rows3, size_t cap_rows);

// hrx_compile.cpp: regex -> minimal DFA in the reference's numbering; either output may be null
struct CompiledDfa {               // Vec<serde_json::Value> of get_dfa_json_value, edges in map (byte) order of the key text
    struct Edge { std::string key; int to; std::vector<uint16_t> syms; };
    struct Node { bool accept; std::vector<Edge> edges; };
    std::vector<Node> nodes;
};
bool compile_regex(const char *regex, size_t len, std::string *json_out, std::string *text_out, std::string &err,
                   CompiledDfa *dfa_out = nullptr);

// hrx_substr.cpp: DecomposedRegexConfig::gen_regex_files (src/vrm/mod.rs:62-307): allstr text + one substr text per public part
struct RegexPart { std::string regex_def; bool is_public; size_t max_size; };
struct RegexFiles { std::string allstr; std::vector<std::string> substrs; };
bool gen_regex_files(const std::vector<RegexPart> &parts, size_t max_byte_size, RegexFiles &out, std::string &err);
std::string format_regex_printable(const std::string &s);                      // formatRegexPrintable, src/vrm/regex.js:24-39
// leftmost-first search (regex crate semantics, which fancy-regex delegates to); false if no match; err set on syntax it does not cover
bool regex_find(const std::string &pattern, const std::string &text, size_t &start, size_t &end, bool &found, std::string &err);

// LDS budget for the table image (the rest of the 160 KiB holds the per-wave staging)
constexpr size_t kMaxTableBytes = 2048 * 1024;  // 2048 table rows; beyond the LDS budget the kernels read the table from global memory

}  // namespace hrx
