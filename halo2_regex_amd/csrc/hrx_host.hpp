// hrx_host.hpp — C++ host-side mirror of the reference's Rust surface for the path, over the C ABI (include/hrx.h).
// Header only.  Same names, argument meaning and error behaviour as src/defs.rs and src/lib.rs:
//   AllstrRegexDef::read_from_text / SubstrRegexDef::read_from_text / ::new     defs.rs:54, 184, 147
//   RegexVerifyConfig::configure / load / derive_states / derive_substr_ids /
//     derive_is_start_end / match_substrs                                       lib.rs:126, 779, 804, 825, 847, 311
// Where the reference panics (parse errors, lib.rs:817) these throw std::runtime_error with the same text.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/hrx.h"

namespace hrx_host {

inline void check(int rc) {
    if (rc != HRX_OK) throw std::runtime_error(hrx_last_error());
}

struct AllstrRegexDef {  // defs.rs:26-36 (parsed inside the library)
    std::string text, path;
    static AllstrRegexDef read_from_text(const std::string &file_path) { AllstrRegexDef d; d.path = file_path; return d; }
    static AllstrRegexDef from_string(std::string t) { AllstrRegexDef d; d.text = std::move(t); return d; }
};

struct SubstrRegexDef {  // defs.rs:115-132
    std::string text, path;
    std::vector<std::pair<uint64_t, uint64_t>> valid_state_transitions;
    std::vector<uint64_t> start_states, end_states;
    bool structured = false;
    static SubstrRegexDef read_from_text(const std::string &file_path) { SubstrRegexDef d; d.path = file_path; return d; }
    static SubstrRegexDef from_string(std::string t) { SubstrRegexDef d; d.text = std::move(t); return d; }
    // SubstrRegexDef::new (defs.rs:147-163); max_length/min_position/max_position are unused by the chip
    static SubstrRegexDef new_(size_t, uint64_t, uint64_t, std::vector<std::pair<uint64_t, uint64_t>> transitions,
                               std::vector<uint64_t> starts, std::vector<uint64_t> ends) {
        SubstrRegexDef d;
        d.structured = true;
        d.valid_state_transitions = std::move(transitions);
        d.start_states = std::move(starts);
        d.end_states = std::move(ends);
        return d;
    }
};

struct RegexDefs {  // defs.rs:17-22
    AllstrRegexDef allstr;
    std::vector<SubstrRegexDef> substrs;
};

// AssignedRegexResult (lib.rs:79-93), integers instead of assigned cells
struct AssignedRegexResult {
    std::vector<uint64_t> all_enable_flags, all_characters, all_substr_ids, masked_characters;
    std::vector<std::vector<uint64_t>> states, substr_ids, start_enables, end_enables;  // [def][row]
    uint64_t status = 0;
};

class RegexVerifyConfig {
   public:
    std::vector<RegexDefs> regex_defs;  // pub field, lib.rs:112

    // configure(meta, max_chars_size, gate, regex_defs) minus the halo2 objects (lib.rs:126-131)
    static RegexVerifyConfig configure(size_t max_chars_size, std::vector<RegexDefs> defs, int device = 0) {
        RegexVerifyConfig c;
        c.max_chars_size_ = max_chars_size;
        c.regex_defs = std::move(defs);
        check(hrx_defs_create(&c.defs_));
        for (const RegexDefs &rd : c.regex_defs) {
            if (!rd.allstr.path.empty()) check(hrx_defs_push_allstr_file(c.defs_, rd.allstr.path.c_str()));
            else check(hrx_defs_push_allstr_text(c.defs_, rd.allstr.text.data(), rd.allstr.text.size()));
            for (const SubstrRegexDef &sd : rd.substrs) {
                if (sd.structured) {
                    std::vector<uint64_t> a, b;
                    for (auto &p : sd.valid_state_transitions) { a.push_back(p.first); b.push_back(p.second); }
                    check(hrx_defs_push_substr(c.defs_, a.size(), a.data(), b.data(), sd.start_states.size(), sd.start_states.data(),
                                               sd.end_states.size(), sd.end_states.data()));
                } else if (!sd.path.empty()) check(hrx_defs_push_substr_file(c.defs_, sd.path.c_str()));
                else check(hrx_defs_push_substr_text(c.defs_, sd.text.data(), sd.text.size()));
            }
        }
        check(hrx_defs_finalize(c.defs_));
        if (device >= 0) check(hrx_ctx_create(c.defs_, device, &c.ctx_));
        return c;
    }
    RegexVerifyConfig() = default;
    RegexVerifyConfig(RegexVerifyConfig &&o) noexcept { *this = std::move(o); }
    RegexVerifyConfig &operator=(RegexVerifyConfig &&o) noexcept {
        std::swap(defs_, o.defs_); std::swap(ctx_, o.ctx_); std::swap(max_chars_size_, o.max_chars_size_);
        regex_defs = std::move(o.regex_defs);
        return *this;
    }
    RegexVerifyConfig(const RegexVerifyConfig &) = delete;
    ~RegexVerifyConfig() { if (ctx_) hrx_ctx_destroy(ctx_); if (defs_) hrx_defs_destroy(defs_); }

    size_t num_defs() const { return hrx_defs_num_defs(defs_); }
    hrx_ctx *ctx() const { return ctx_; }

    // load (lib.rs:779-785): the integer rows RegexTableConfig::load assigns, per def
    std::vector<std::vector<uint64_t>> load_transition_rows() const {
        std::vector<std::vector<uint64_t>> out;
        for (size_t d = 0; d < num_defs(); ++d) {
            std::vector<uint64_t> r(4 * hrx_table_transition_rows(defs_, d, nullptr, 0));
            hrx_table_transition_rows(defs_, d, r.data(), r.size() / 4);
            out.push_back(std::move(r));
        }
        return out;
    }

    std::vector<std::vector<uint64_t>> derive_states(const std::vector<uint8_t> &characters) const {  // lib.rs:804
        const size_t n = characters.size(), D = num_defs();
        std::vector<uint64_t> flat(D * (n + 1));
        check(hrx_derive_states(ctx_, characters.data(), n, flat.data()));
        return split(flat, D, n + 1);
    }
    std::vector<std::vector<uint64_t>> derive_substr_ids(const std::vector<std::vector<uint64_t>> &states) const {  // lib.rs:825
        const size_t D = states.size(), n = states[0].size() - 1;
        std::vector<uint64_t> out(D * n);
        check(hrx_derive_substr_ids(ctx_, join(states).data(), n, out.data()));
        return split(out, D, n);
    }
    std::pair<std::vector<std::vector<uint8_t>>, std::vector<std::vector<uint8_t>>> derive_is_start_end(
        const std::vector<std::vector<uint64_t>> &states, const std::vector<std::vector<uint64_t>> &substr_ids) const {  // lib.rs:847
        const size_t D = states.size(), n = states[0].size() - 1;
        std::vector<uint8_t> st(D * (n + 1)), en(D * (n + 1));
        check(hrx_derive_is_start_end(ctx_, join(states).data(), join(substr_ids).data(), n, st.data(), en.data()));
        std::vector<std::vector<uint8_t>> a(D), b(D);
        for (size_t d = 0; d < D; ++d) {
            a[d].assign(st.begin() + d * (n + 1), st.begin() + (d + 1) * (n + 1));
            b[d].assign(en.begin() + d * (n + 1), en.begin() + (d + 1) * (n + 1));
        }
        return {a, b};
    }
    AssignedRegexResult match_substrs(const std::vector<uint8_t> &characters) const {  // lib.rs:311
        const size_t M = max_chars_size_, D = num_defs();
        AssignedRegexResult r;
        r.all_enable_flags.resize(M); r.all_characters.resize(M); r.all_substr_ids.resize(M); r.masked_characters.resize(M);
        std::vector<uint64_t> st(D * M), sid(D * M), se(D * M), ee(D * M);
        check(hrx_match_substrs(ctx_, characters.data(), characters.size(), M, r.all_enable_flags.data(), r.all_characters.data(),
                                st.data(), sid.data(), se.data(), ee.data(), r.masked_characters.data(), r.all_substr_ids.data(),
                                &r.status));
        r.states = split(st, D, M); r.substr_ids = split(sid, D, M); r.start_enables = split(se, D, M); r.end_enables = split(ee, D, M);
        return r;
    }

   private:
    static std::vector<std::vector<uint64_t>> split(const std::vector<uint64_t> &f, size_t D, size_t len) {
        std::vector<std::vector<uint64_t>> o(D);
        for (size_t d = 0; d < D; ++d) o[d].assign(f.begin() + d * len, f.begin() + (d + 1) * len);
        return o;
    }
    static std::vector<uint64_t> join(const std::vector<std::vector<uint64_t>> &v) {
        std::vector<uint64_t> o;
        for (auto &x : v) o.insert(o.end(), x.begin(), x.end());
        return o;
    }
    hrx_defs *defs_ = nullptr;
    hrx_ctx *ctx_ = nullptr;
    size_t max_chars_size_ = 0;
};

}  // namespace hrx_host
