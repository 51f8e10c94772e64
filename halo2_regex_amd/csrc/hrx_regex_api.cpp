// hrx_regex_api.cpp — the C ABI of definition generation (hrx_compile.cpp, hrx_substr.cpp): host only, no context.
#include "hrx_ctx.hpp"

using namespace hrx;

// ---------------------------------------------------------------- definition generation (hrx_compile.cpp)
static int regex_compile_entry(const char *regex, size_t regex_len, char *out, size_t cap, size_t *needed, bool json) {
    if ((!regex && regex_len) || (!out && cap) || !needed) return fail(HRX_ERR_ARG, "null argument");
    std::string res, err;
    bool ok = json ? hrx::compile_regex(regex, regex_len, &res, nullptr, err) : hrx::compile_regex(regex, regex_len, nullptr, &res, err);
    if (!ok) return fail(HRX_ERR_PARSE, err);
    *needed = res.size();
    if (out) memcpy(out, res.data(), std::min(cap, res.size()));
    return HRX_OK;
}
extern "C" int hrx_regex_to_allstr_text(const char *regex, size_t regex_len, char *out, size_t cap, size_t *needed) {
    return regex_compile_entry(regex, regex_len, out, cap, needed, false);
}
extern "C" int hrx_regex_to_dfa_json(const char *regex, size_t regex_len, char *out, size_t cap, size_t *needed) {
    return regex_compile_entry(regex, regex_len, out, cap, needed, true);
}

struct hrx_regex_files { hrx::RegexFiles f; };
extern "C" int hrx_gen_regex_files(const hrx_regex_part *parts, size_t n_parts, size_t max_byte_size, hrx_regex_files **out) {
    if ((!parts && n_parts) || !out) return fail(HRX_ERR_ARG, "null argument");
    if (max_byte_size == 0) return fail(HRX_ERR_ARG, "max_byte_size must be positive");
    std::vector<hrx::RegexPart> ps;
    for (size_t i = 0; i < n_parts; i++) {
        if (!parts[i].regex_def && parts[i].regex_len) return fail(HRX_ERR_ARG, "null regex_def");
        ps.push_back({std::string(parts[i].regex_def ? parts[i].regex_def : "", parts[i].regex_len), parts[i].is_public != 0, parts[i].max_size});
    }
    auto *res = new hrx_regex_files();
    std::string err;
    if (!hrx::gen_regex_files(ps, max_byte_size, res->f, err)) { delete res; return fail(HRX_ERR_PARSE, err); }
    *out = res;
    return HRX_OK;
}
extern "C" size_t hrx_regex_files_num_substrs(const hrx_regex_files *files) { return files ? files->f.substrs.size() : 0; }
extern "C" const char *hrx_regex_files_allstr(const hrx_regex_files *files, size_t *len) {
    if (!files) return nullptr;
    if (len) *len = files->f.allstr.size();
    return files->f.allstr.data();
}
extern "C" const char *hrx_regex_files_substr(const hrx_regex_files *files, size_t idx, size_t *len) {
    if (!files || idx >= files->f.substrs.size()) return nullptr;
    if (len) *len = files->f.substrs[idx].size();
    return files->f.substrs[idx].data();
}
extern "C" void hrx_regex_files_destroy(hrx_regex_files *files) { delete files; }
extern "C" int hrx_format_regex_str(const char *regex, size_t regex_len, char *out, size_t cap, size_t *needed) {
    if ((!regex && regex_len) || (!out && cap) || !needed) return fail(HRX_ERR_ARG, "null argument");
    std::string res = hrx::format_regex_printable(std::string(regex ? regex : "", regex_len));
    *needed = res.size();
    if (out) memcpy(out, res.data(), std::min(cap, res.size()));
    return HRX_OK;
}
extern "C" int hrx_regex_find(const char *pattern, size_t pattern_len, const char *text, size_t text_len, int *found, size_t *start,
                              size_t *end) {
    if ((!pattern && pattern_len) || (!text && text_len) || !found || !start || !end) return fail(HRX_ERR_ARG, "null argument");
    std::string err;
    bool f = false;
    size_t s = 0, e = 0;
    if (!hrx::regex_find(std::string(pattern ? pattern : "", pattern_len), std::string(text ? text : "", text_len), s, e, f, err))
        return fail(HRX_ERR_PARSE, err);
    *found = f; *start = s; *end = e;
    return HRX_OK;
}
