// hrx_substr.cpp — SubstrRegexDef generation: the second half of
// DecomposedRegexConfig::gen_regex_files (src/vrm/mod.rs:62-307), i.e. extract_substr_ids (mod.rs:309-538) and
// get_substr_defs_from_path (mod.rs:540-600), natively and without V8 / fancy-regex.
//
// What the reference does, restated:
//   1. the minimal DFA of the concatenated parts (hrx_compile.cpp) becomes a graph with REVERSED edges, one per
//      (state, next) pair, labelled with the concatenated symbols of the edge key (js_caller.rs:88-125);
//   2. every simple path accept -> ... -> 0 of that graph is enumerated; states popped on the way that carry a
//      self-loop are remembered (`self_nodes`) (mod.rs:355-387);
//   3. each path is spelled as a representative string — the first symbol of every edge label — and the cumulative
//      regexes part_0, part_0 part_1, ... (each part passed through formatRegexPrintable, regex.js:24-39) are searched
//      in it; the match ends cut the path into the state ranges of the parts (mod.rs:389-397, 540-600);
//   4. for every public part: first/last state of its range are its start/end states; consecutive pairs, self-loops of
//      range states, and back edges from a later to an earlier state of the range are its valid transitions
//      (mod.rs:452-496);
//   5. the text format of mod.rs:268-304 (sorted endpoints with a trailing blank, pairs sorted by (cur, next)).
//
// Third-party piece: fancy-regex 0.11 (Cargo.toml:22), absent from /root/reference.  For patterns without look-around
// or back-references it delegates to the `regex` crate, whose documented search semantics are leftmost-first
// (Perl-like priorities: earlier alternative first, greedy quantifiers prefer more).  regex_find() below restates
// that as a priority-ordered Thompson simulation (Pike VM) over the syntax subset formatRegexPrintable can emit from
// the DFA dialect: literals, escapes, groups, | * + ? {m,n} (and lazy forms), `.`, bracket classes.  Anything else
// is rejected with HRX_ERR_PARSE rather than guessed.  Parity: pinned by the reference's committed substr*.txt
// files (tests/test_substr_gen.py) and, for the matcher alone, cross-checked against CPython's `re` on that subset.
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "hrx_defs.hpp"

namespace hrx {

// ---------------------------------------------------------------- formatRegexPrintable (regex.js:24-39)
static void replace_all(std::string &s, const std::string &from, const std::string &to) {
    if (from.empty()) return;
    std::string out;
    size_t i = 0;
    for (;;) {
        size_t f = s.find(from, i);
        if (f == std::string::npos) break;
        out.append(s, i, f - i);
        out += to;
        i = f + from.size();
    }
    out.append(s, i, std::string::npos);
    s.swap(out);
}

std::string format_regex_printable(const std::string &in) {
    std::string s;                                 // JSON.stringify(s) without the surrounding quotes
    char buf[8];
    for (unsigned char c : in) {
        switch (c) {
        case '"': s += "\\\""; break;
        case '\\': s += "\\\\"; break;
        case '\b': s += "\\b"; break;
        case '\f': s += "\\f"; break;
        case '\n': s += "\\n"; break;
        case '\r': s += "\\r"; break;
        case '\t': s += "\\t"; break;
        default:
            if (c < 0x20) { snprintf(buf, sizeof buf, "\\u%04x", c); s += buf; }
            else s += (char)c;
        }
    }
    replace_all(s, "\\\\\\\\", "\\");
    replace_all(s, "\\\\", "\\");
    replace_all(s, "/", "\\/");
    replace_all(s, "\x0b", "\\\xe2\x99\xa5");       // never fires: stringify already wrote U+000B as \\u000b
    replace_all(s, "^", "\\^");
    replace_all(s, "$", "\\$");
    replace_all(s, "|[|", "|\\[|");
    replace_all(s, "|]|", "|\\]|");
    replace_all(s, "|.|", "|\\.|");
    replace_all(s, "|$|", "|\\$|");
    replace_all(s, "|^|", "|\\^|");
    return s;
}

// ---------------------------------------------------------------- leftmost-first regex search
namespace re {

using ByteSet = std::array<uint64_t, 4>;
static void bs_add(ByteSet &b, unsigned c) { b[c >> 6] |= 1ull << (c & 63); }
static bool bs_has(const ByteSet &b, unsigned c) { return b[c >> 6] >> (c & 63) & 1; }
static void bs_range(ByteSet &b, unsigned lo, unsigned hi) { for (unsigned c = lo; c <= hi; c++) bs_add(b, c); }
static void bs_not(ByteSet &b) { for (auto &w : b) w = ~w; }

enum Kind { K_EMPTY, K_SET, K_CAT, K_ALT, K_REP };
struct Node { Kind kind; ByteSet set{}; std::vector<int> kids; int lo = 0, hi = 0; bool greedy = true; };   // hi < 0: unbounded

struct Parser {
    const std::string &p;
    size_t i = 0;
    std::vector<Node> nodes;
    std::string err;
    explicit Parser(const std::string &pat) : p(pat) {}

    int add(Node n) { nodes.push_back(std::move(n)); return (int)nodes.size() - 1; }
    int lit(unsigned c) { Node n; n.kind = K_SET; bs_add(n.set, c); return add(n); }
    bool fail(const std::string &m) { if (err.empty()) err = "regex syntax outside the supported subset at " + std::to_string(i) + ": " + m; return false; }

    static constexpr uint32_t kRawByte = 0x80000000u;
    static int hexv(char c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; }

    static void perl_class(char c, ByteSet &b) {
        ByteSet t{};
        switch (c | 0x20) {
        case 'd': bs_range(t, '0', '9'); break;
        case 'w': bs_range(t, '0', '9'); bs_range(t, 'a', 'z'); bs_range(t, 'A', 'Z'); bs_add(t, '_'); break;
        case 's': for (unsigned x : {9u, 10u, 11u, 12u, 13u, 32u}) bs_add(t, x); break;
        }
        if (c >= 'A' && c <= 'Z') bs_not(t);
        for (int k = 0; k < 4; k++) b[k] |= t[k];
    }

    // after the backslash; returns false on error; is_class: filled `cls`; else code point in cp
    bool escape(uint32_t &cp, bool &is_class, ByteSet &cls) {
        if (i >= p.size()) return fail("trailing backslash");
        char c = p[i++];
        is_class = false;
        switch (c) {
        case 'n': cp = '\n'; return true;
        case 'r': cp = '\r'; return true;
        case 't': cp = '\t'; return true;
        case 'f': cp = '\f'; return true;
        case 'v': cp = '\v'; return true;
        case 'a': cp = 7; return true;
        case 'd': case 'w': case 's': case 'D': case 'W': case 'S': is_class = true; perl_class(c, cls); return true;
        case 'x': case 'u': case 'U': {
            size_t want = c == 'x' ? 2 : c == 'u' ? 4 : 8;
            cp = 0;
            if (i < p.size() && p[i] == '{') {
                size_t j = i + 1, digits = 0;
                while (j < p.size() && hexv(p[j]) >= 0) { cp = cp * 16 + hexv(p[j]); j++; digits++; }
                if (j >= p.size() || p[j] != '}' || !digits) return fail("bad \\x{..}");
                i = j + 1;
                return true;
            }
            for (size_t k = 0; k < want; k++) {
                if (i >= p.size() || hexv(p[i]) < 0) return fail("bad hex escape");
                cp = cp * 16 + hexv(p[i++]);
            }
            return true;
        }
        }
        if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9')) return fail(std::string("escape \\") + c);
        cp = (unsigned char)c;                       // escaped punctuation; a byte >= 0x80 is the lead byte of an escaped
        if (cp >= 0x80) cp |= kRawByte;              // multi-byte character: matched as that byte, the rest follow as literals
        return true;
    }

    int literal_cp(uint32_t cp) {                    // code point -> its UTF-8 bytes in sequence
        if (cp & kRawByte) return lit(cp & 0xff);
        if (cp < 0x80) return lit(cp);
        std::string u;
        if (cp < 0x800) { u += (char)(0xc0 | cp >> 6); u += (char)(0x80 | (cp & 0x3f)); }
        else if (cp < 0x10000) { u += (char)(0xe0 | cp >> 12); u += (char)(0x80 | ((cp >> 6) & 0x3f)); u += (char)(0x80 | (cp & 0x3f)); }
        else { u += (char)(0xf0 | cp >> 18); u += (char)(0x80 | ((cp >> 12) & 0x3f)); u += (char)(0x80 | ((cp >> 6) & 0x3f)); u += (char)(0x80 | (cp & 0x3f)); }
        Node n; n.kind = K_CAT;
        for (unsigned char b : u) n.kids.push_back(lit(b));
        return add(n);
    }

    int parse_class() {                              // after '['
        Node n; n.kind = K_SET;
        bool neg = false;
        if (i < p.size() && p[i] == '^') { neg = true; i++; }
        bool first = true;
        for (;;) {
            if (i >= p.size()) { fail("unclosed ["); return -1; }
            unsigned char c = p[i];
            if (c == ']' && !first) { i++; break; }
            first = false;
            uint32_t lo;
            if (c == '[') { fail("nested class"); return -1; }
            if (c == '\\') {
                i++;
                bool is_class; ByteSet cls{};
                if (!escape(lo, is_class, cls)) return -1;
                if (is_class) { for (int k = 0; k < 4; k++) n.set[k] |= cls[k]; continue; }
            } else { lo = c; i++; }
            if (lo >= 0x80) { fail("non-ASCII in class"); return -1; }   // also covers kRawByte
            uint32_t hi = lo;
            if (i + 1 < p.size() && p[i] == '-' && p[i + 1] != ']') {
                i++;
                unsigned char d = p[i];
                if (d == '\\') {
                    i++;
                    bool is_class; ByteSet cls{};
                    if (!escape(hi, is_class, cls)) return -1;
                    if (is_class) { fail("class as range end"); return -1; }
                } else { hi = d; i++; }
                if (hi >= 0x80 || hi < lo) { fail("bad range"); return -1; }
            }
            bs_range(n.set, lo, hi);
        }
        if (neg) bs_not(n.set);
        return add(n);
    }

    bool parse_counts(size_t at, int &lo, int &hi, size_t &next) {   // p[at] == '{'; valid {m} {m,} {m,n}?
        size_t j = at + 1;
        auto num = [&](int &v) { size_t s = j; long x = 0; while (j < p.size() && p[j] >= '0' && p[j] <= '9' && j - s < 6) x = x * 10 + (p[j++] - '0'); v = (int)x; return j > s; };
        if (!num(lo)) return false;
        hi = lo;
        if (j < p.size() && p[j] == ',') {
            j++;
            if (j < p.size() && p[j] == '}') hi = -1;
            else if (!num(hi)) return false;
        }
        if (j >= p.size() || p[j] != '}') return false;
        if (hi >= 0 && hi < lo) return false;
        next = j + 1;
        return true;
    }

    int parse_atom() {
        unsigned char c = p[i];
        if (c == '(') {
            i++;
            if (i + 1 < p.size() && p[i] == '?') {
                if (p[i + 1] == ':') i += 2;
                else { fail("group flag / look-around"); return -1; }
            }
            int a = parse_alt();
            if (a < 0) return -1;
            if (i >= p.size() || p[i] != ')') { fail("missing )"); return -1; }
            i++;
            return a;
        }
        if (c == '[') { i++; return parse_class(); }
        if (c == '.') { i++; Node n; n.kind = K_SET; bs_range(n.set, 0, 255); n.set[0] &= ~(1ull << '\n'); return add(n); }
        if (c == '\\') {
            i++;
            uint32_t cp; bool is_class; ByteSet cls{};
            if (!escape(cp, is_class, cls)) return -1;
            if (is_class) { Node n; n.kind = K_SET; n.set = cls; return add(n); }
            return literal_cp(cp);
        }
        if (c == '^' || c == '$') { fail("anchor"); return -1; }
        if (c == '*' || c == '+' || c == '?') { fail("repetition operator missing expression"); return -1; }
        i++;
        return lit(c);                               // includes ] { } and the bytes of non-ASCII characters
    }

    int parse_repeat() {
        int a = parse_atom();
        if (a < 0) return -1;
        while (i < p.size()) {
            char c = p[i];
            int lo, hi;
            size_t next;
            if (c == '*') { lo = 0; hi = -1; next = i + 1; }
            else if (c == '+') { lo = 1; hi = -1; next = i + 1; }
            else if (c == '?') { lo = 0; hi = 1; next = i + 1; }
            else if (c == '{' && parse_counts(i, lo, hi, next)) {}
            else break;
            i = next;
            Node n; n.kind = K_REP; n.kids = {a}; n.lo = lo; n.hi = hi;
            if (i < p.size() && p[i] == '?') { n.greedy = false; i++; }
            if (lo > 1000 || hi > 1000) { fail("repeat count"); return -1; }
            a = add(n);
        }
        return a;
    }

    int parse_cat() {
        Node n; n.kind = K_CAT;
        while (i < p.size() && p[i] != '|' && p[i] != ')') {
            int r = parse_repeat();
            if (r < 0) return -1;
            n.kids.push_back(r);
        }
        if (n.kids.empty()) { Node e; e.kind = K_EMPTY; return add(e); }
        if (n.kids.size() == 1) return n.kids[0];
        return add(n);
    }

    int parse_alt() {
        Node n; n.kind = K_ALT;
        for (;;) {
            int c = parse_cat();
            if (c < 0) return -1;
            n.kids.push_back(c);
            if (i < p.size() && p[i] == '|') { i++; continue; }
            break;
        }
        if (n.kids.size() == 1) return n.kids[0];
        return add(n);
    }
};

enum Op { O_SET, O_SPLIT, O_JMP, O_MATCH };
struct Inst { Op op; int x = 0, y = 0; ByteSet set{}; };

struct Compiler {
    const std::vector<Node> &nodes;
    std::vector<Inst> prog;
    explicit Compiler(const std::vector<Node> &n) : nodes(n) {}
    int emit(Inst i) { prog.push_back(i); return (int)prog.size() - 1; }

    void gen(int id) {
        const Node &n = nodes[id];
        switch (n.kind) {
        case K_EMPTY: break;
        case K_SET: { Inst i; i.op = O_SET; i.set = n.set; emit(i); break; }
        case K_CAT: for (int k : n.kids) gen(k); break;
        case K_ALT: {
            std::vector<int> jumps;
            for (size_t k = 0; k < n.kids.size(); k++) {
                if (k + 1 < n.kids.size()) {
                    int sp = emit({O_SPLIT});
                    prog[sp].x = sp + 1;
                    gen(n.kids[k]);
                    jumps.push_back(emit({O_JMP}));
                    prog[sp].y = (int)prog.size();
                } else gen(n.kids[k]);
            }
            for (int j : jumps) prog[j].x = (int)prog.size();
            break;
        }
        case K_REP: {
            for (int k = 0; k < n.lo; k++) gen(n.kids[0]);
            if (n.hi < 0) {                          // e*: L: split body, out; body; jmp L
                int sp = emit({O_SPLIT});
                gen(n.kids[0]);
                int j = emit({O_JMP});
                prog[j].x = sp;
                int out = (int)prog.size();
                if (n.greedy) { prog[sp].x = sp + 1; prog[sp].y = out; } else { prog[sp].x = out; prog[sp].y = sp + 1; }
            } else {                                 // (e(e(e)?)?)? for the optional copies
                std::vector<int> splits;
                for (int k = n.lo; k < n.hi; k++) {
                    splits.push_back(emit({O_SPLIT}));
                    gen(n.kids[0]);
                }
                int out = (int)prog.size();
                for (int sp : splits) { if (n.greedy) { prog[sp].x = sp + 1; prog[sp].y = out; } else { prog[sp].x = out; prog[sp].y = sp + 1; } }
            }
            break;
        }
        }
    }
};

struct Vm {
    const std::vector<Inst> &prog;
    std::vector<int> mark;
    int gen_id = 0;
    explicit Vm(const std::vector<Inst> &p) : prog(p), mark(p.size(), -1) {}

    void add(std::vector<int> &list, int pc) {      // follow jumps / splits in priority order
        if (mark[pc] == gen_id) return;
        mark[pc] = gen_id;
        const Inst &in = prog[pc];
        if (in.op == O_JMP) add(list, in.x);
        else if (in.op == O_SPLIT) { add(list, in.x); add(list, in.y); }
        else list.push_back(pc);
    }

    // anchored at `from`; returns the end of the leftmost-first match or -1
    long run(const std::string &text, size_t from) {
        std::vector<int> clist, nlist;
        long matched = -1;
        gen_id++;
        add(clist, 0);
        for (size_t pos = from;; pos++) {
            if (clist.empty()) break;
            gen_id++;
            nlist.clear();
            for (int pc : clist) {
                const Inst &in = prog[pc];
                if (in.op == O_MATCH) { matched = (long)pos; break; }           // lower-priority threads are cut
                if (pos < text.size() && bs_has(in.set, (unsigned char)text[pos])) add(nlist, pc + 1);
            }
            if (pos >= text.size()) break;
            clist.swap(nlist);
        }
        return matched;
    }
};

}  // namespace re

bool regex_find(const std::string &pattern, const std::string &text, size_t &start, size_t &end, bool &found, std::string &err) {
    re::Parser ps(pattern);
    int root = ps.parse_alt();
    if (root >= 0 && ps.i < pattern.size()) { ps.fail("unmatched )"); root = -1; }
    if (root < 0) { err = ps.err; return false; }
    re::Compiler c(ps.nodes);
    c.gen(root);
    c.emit({re::O_MATCH});
    re::Vm vm(c.prog);
    found = false;
    for (size_t s = 0; s <= text.size(); s++) {
        long e = vm.run(text, s);
        if (e >= 0) { found = true; start = s; end = (size_t)e; break; }
    }
    return true;
}

// ---------------------------------------------------------------- extract_substr_ids + file text
bool gen_regex_files(const std::vector<RegexPart> &parts, size_t max_byte_size, RegexFiles &out, std::string &err) {
    std::string all_regex;
    for (auto &p : parts) all_regex += p.regex_def;                              // mod.rs:87-91
    CompiledDfa dfa;
    if (!compile_regex(all_regex.data(), all_regex.size(), nullptr, &out.allstr, err, &dfa)) return false;
    out.substrs.clear();

    // graph (js_caller.rs:88-125): one labelled edge per (state, next); max_state = largest target (js_caller.rs:67-86)
    size_t max_state = 0;
    long accepted = -1;
    for (size_t i = 0; i < dfa.nodes.size(); i++) {
        if (accepted < 0 && dfa.nodes[i].accept) accepted = (long)i;
        for (auto &e : dfa.nodes[i].edges) max_state = std::max(max_state, (size_t)e.to);
    }
    if (accepted < 0) { err = "No accepted state"; return false; }
    const size_t N = max_state + 1;
    if ((size_t)accepted >= N) { err = "accepted state has no incoming edge"; return false; }
    std::vector<std::map<int, std::string>> label(N);        // label[from][to]
    std::vector<std::vector<int>> preds(N);
    for (size_t i = 0; i < dfa.nodes.size() && i < N; i++)
        for (auto &e : dfa.nodes[i].edges) {
            std::string s;
            for (uint16_t c : e.syms) {
                if (c >= 0x80) { err = "substring definitions need single-byte symbols (js_caller.rs:120 asserts key_char.len() == 1)"; return false; }
                s += (char)c;
            }
            label[i][e.to] = s;
            preds[e.to].push_back((int)i);
        }

    // simple paths accept -> 0 along reversed edges (mod.rs:355-387)
    std::set<int> self_nodes;
    std::vector<std::vector<int>> pathes;
    {
        std::vector<std::pair<int, std::vector<int>>> stack;
        stack.push_back({(int)accepted, {(int)accepted}});
        size_t work = 0;
        while (!stack.empty()) {
            auto top = std::move(stack.back());
            stack.pop_back();
            const int node = top.first;
            const std::vector<int> &path = top.second;
            for (int parent : preds[node]) {
                if (parent == node) { self_nodes.insert(node); continue; }
                if (std::find(path.begin(), path.end(), parent) != path.end()) continue;
                if (parent == 0) { pathes.push_back(path); continue; }
                std::vector<int> np = path;
                np.push_back(parent);
                stack.push_back({parent, std::move(np)});
                if (++work > 20000000) { err = "too many simple paths through the DFA"; return false; }
            }
        }
    }

    // cumulative part regexes (mod.rs:389-405)
    std::vector<size_t> public_idx;
    std::vector<std::string> part_regex;
    for (size_t i = 0; i < parts.size(); i++) {
        if (parts[i].is_public) public_idx.push_back(i);
        part_regex.push_back((i ? part_regex[i - 1] : std::string()) + format_regex_printable(parts[i].regex_def));
    }
    std::vector<std::set<std::pair<int, int>>> defs(public_idx.size());
    std::vector<std::set<int>> starts(public_idx.size()), ends(public_idx.size());

    for (auto &p : pathes) {
        std::vector<int> states{0};
        states.insert(states.end(), p.rbegin(), p.rend());                       // 0 ... accepted
        std::string concat;
        for (size_t k = 0; k + 1 < states.size(); k++) {
            auto f = label[states[k]].find(states[k + 1]);
            if (f == label[states[k]].end() || f->second.empty()) { err = "No edge in the graph"; return false; }
            concat += f->second[0];                                              // mod.rs:548-552
        }
        std::vector<size_t> index_ends;
        for (auto &rgx : part_regex) {                                           // mod.rs:553-583
            size_t s = 0, e = 0; bool found = false;
            if (!regex_find(rgx, concat, s, e, found, err)) return false;
            if (!found) { err = "a part regex does not match a path of the DFA (the reference unwraps None here)"; return false; }
            index_ends.push_back(s == e ? e + 1 : e);
        }
        for (size_t si = 0; si < public_idx.size(); si++) {                      // mod.rs:584-598, 452-496
            const size_t idx = public_idx[si];
            const size_t start = idx == 0 ? 0 : index_ends[idx - 1], end = index_ends[idx];
            if (end >= states.size() || start > end) { err = "part boundaries fall outside the path (the reference's slice would panic)"; return false; }
            std::vector<int> ps(states.begin() + start, states.begin() + end + 1);
            starts[si].insert(ps.front());
            ends[si].insert(ps.back());
            for (size_t k = 0; k + 1 < ps.size(); k++) {
                defs[si].insert({ps[k], ps[k + 1]});
                if (self_nodes.count(ps[k])) defs[si].insert({ps[k], ps[k]});
                for (size_t pre = 0; pre <= k; pre++)                            // DFA edge ps[k+1] -> ps[pre]
                    if (label[ps[k + 1]].count(ps[pre])) defs[si].insert({ps[k + 1], ps[pre]});
            }
            // mod.rs:483-495: the self-loop of the last state is kept when the part regex still matches the path's
            // string extended by the loop's symbol — an unanchored search of a string that already contains a match,
            // so it always does.
            if (self_nodes.count(ps.back())) defs[si].insert({ps.back(), ps.back()});
        }
    }

    for (size_t si = 0; si < public_idx.size(); si++) {                          // mod.rs:268-304
        std::string t = std::to_string(parts[public_idx[si]].max_size) + "\n0\n" + std::to_string(max_byte_size - 1) + "\n";
        for (int s : starts[si]) t += std::to_string(s) + " ";
        t += "\n";
        for (int e : ends[si]) t += std::to_string(e) + " ";
        t += "\n";
        for (auto &d : defs[si]) t += std::to_string(d.first) + " " + std::to_string(d.second) + "\n";
        out.substrs.push_back(t);
    }
    return true;
}

}  // namespace hrx
