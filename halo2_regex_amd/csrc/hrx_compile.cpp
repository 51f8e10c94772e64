// hrx_compile.cpp — regex -> minimal DFA -> AllstrRegexDef text, natively.
//
// Replaces, for the table step that feeds the witness path, the reference's V8 round trip:
//   DecomposedRegexConfig::gen_regex_files            src/vrm/mod.rs:62-95      (concatenate the parts' regex_def)
//   get_dfa_json_value -> regexToDfa                  src/vrm/js_caller.rs:43-48, src/vrm/regex.js:40-92
//   parseRegex / regexToNfa / nfaToDfa / minDfa       src/vrm/regex.js:236-367, 375-437, 445-551, 559-762
//   dfa_to_regex_def_text                             src/vrm/js_caller.rs:127-157
// The output has to be byte-identical to what that pipeline writes, because state numbers are part of the circuit's
// fixed tables and of every SubstrRegexDef that names them.  So this file keeps the observable ORDER rules of the
// pipeline, not just its language:
//   * the Thompson construction's shape (one NFA per AST visit; `x+` visits x twice) — it decides how many subset
//     states exist before minimisation and therefore their names;
//   * subset states are named A, B, ..., Z, AA, ... in BFS discovery order over code-unit-sorted symbols
//     (regex.js:516-550);
//   * Hopcroft's refinement is run with the reference's queue discipline over JavaScript-object key order — integer-
//     like keys ('0'..'9' as symbols) first in ascending order, then insertion order (regex.js:598-690);
//   * classes are numbered by the string order of their comma-joined member names, the start class swapped to 0
//     (regex.js:698-718); edge keys are the JSON text of the code-unit-sorted symbol list (regex.js:746-752);
//   * the text lists states in index order, edge keys in byte order of the key text (serde_json's Map is a BTreeMap),
//     symbols in key order, each as `cur next (char as u8)`; accepted = first accept state; max = largest target.
// Strings are handled as UTF-16 code units like the JS does; input is UTF-8.
//
// Deviation (documented in DESIGN.md): a pattern ending in a lone backslash is rejected here; the JS reads past the
// end of the string and produces an edge labelled with the text "undefined".
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "hrx_defs.hpp"

namespace hrx {
namespace rx {

using sym_t = int32_t;              // UTF-16 code unit, or kEps
static const sym_t kEps = -1;
static const uint16_t kEpsilonChar = 0x03F5;   // 'ϵ' written in a pattern means the empty string (regex.js:343-346)

// ---------------------------------------------------------------- tokens (regex.js:353-365)
struct Tok { uint16_t ch; bool lit; };   // lit: came from a backslash escape (the JS wraps those in an array)

static bool utf8_to_units(const char *s, size_t n, std::vector<uint16_t> &out, std::string &err) {
    size_t i = 0;
    while (i < n) {
        uint32_t c = (uint8_t)s[i];
        int extra = c < 0x80 ? 0 : (c >> 5) == 6 ? 1 : (c >> 4) == 14 ? 2 : (c >> 3) == 30 ? 3 : -1;
        if (extra < 0 || i + (size_t)extra >= n) { err = "regex is not valid UTF-8"; return false; }
        uint32_t cp = extra == 0 ? c : c & (0x3f >> extra);
        for (int k = 1; k <= extra; k++) {
            uint8_t b = (uint8_t)s[i + k];
            if ((b & 0xc0) != 0x80) { err = "regex is not valid UTF-8"; return false; }
            cp = cp << 6 | (b & 0x3f);
        }
        i += 1 + extra;
        if (cp >= 0x10000) { cp -= 0x10000; out.push_back((uint16_t)(0xd800 | cp >> 10)); out.push_back((uint16_t)(0xdc00 | (cp & 0x3ff))); }
        else out.push_back((uint16_t)cp);
    }
    return true;
}

// ---------------------------------------------------------------- syntax tree (regex.js:236-352)
enum NodeType { N_EMPTY, N_TEXT, N_CAT, N_OR, N_STAR };
struct Ast { NodeType type; uint16_t ch = 0; std::vector<int> parts; int sub = -1; };

// Recursive descent over the token string: alternation -> concatenation -> postfix -> atom.  What the reference's compiler defines (regex.js:236-352) and the
// golden vectors pin (tests/golden/compiler/cases.json, produced by the reference's own script under node) is the TREE — the DFA's state numbers follow its shape:
//   * an alternation of one alternative is that alternative, a concatenation of one factor is that factor;
//   * x+ is cat(x, star(x)) with x's subtree shared (the NFA builder visits it twice), x? is or(x, empty);
//   * a bar splits the scope it is in only while the brackets seen since the scope's start balance: a ')' without a '(' is an ordinary character, and so
//     is every bar behind it in that scope ("a)|b" is the four-character string);
//   * an empty scope — "", "a|", "()" — and a postfix operator without an operand are errors; so is a '(' whose ')' is missing.
// depth[i] = brackets open in front of token i (negative behind a stray ')'): one pass up front, then every scope is a token range.
struct Parser {
    std::vector<Ast> nodes;
    std::string err;
    // Two levels of recursion per bracket level.  The reference's JS throws a catchable RangeError when its own stack runs out (V8: around 10^4 levels); this
    // library promises integer status returns and nothing unwinding (include/hrx.h), so nesting beyond kMaxNesting levels is refused as a parse error.
    static constexpr int kMaxNesting = 2000;
    const std::vector<Tok> *toks = nullptr;
    std::vector<long> depth;

    int add(const Ast &a) { nodes.push_back(a); return (int)nodes.size() - 1; }
    bool raw(size_t i, char c) const { const Tok &k = (*toks)[i]; return !k.lit && k.ch == (uint16_t)(uint8_t)c; }
    int fail(const std::string &what, size_t at) { err = "Error: " + what + " at " + std::to_string(at) + "."; return -1; }

    // the whole expression; returns the root's node index or -1 with err set
    int parse(const std::vector<Tok> &t) {
        toks = &t;
        depth.assign(t.size() + 1, 0);
        for (size_t i = 0; i < t.size(); ++i) depth[i + 1] = depth[i] + (raw(i, '(') ? 1 : raw(i, ')') ? -1 : 0);
        return alternation(0, t.size(), 0);
    }

    // scope [lo, hi): alternatives separated by the bars at the scope's own bracket depth
    int alternation(const size_t lo, const size_t hi, const int nest) {
        if (lo == hi) return fail("empty input", lo);
        if (nest > kMaxNesting) return fail("brackets nested deeper than " + std::to_string(kMaxNesting) + " levels", lo);
        std::vector<int> alts;
        size_t from = lo;
        int result = -2;
        for (size_t i = lo; i <= hi && result == -2; ++i) {
            if (i < hi && !(raw(i, '|') && depth[i] == depth[lo])) continue;
            if (from == lo && i == hi) { result = concatenation(lo, hi, nest); break; }      // no bar: the scope is one concatenation
            const int a = from == i ? fail("empty input", from) : alternation(from, i, nest);   // (an alternative holds no bar of this depth: it comes back as a concatenation)
            if (a < 0) { result = -1; break; }
            alts.push_back(a);
            from = i + 1;
        }
        if (result != -2) return result;
        Ast n; n.type = N_OR; n.parts = alts;
        return add(n);
    }

    // factors of [lo, hi), each an atom with its postfix operators applied as they come
    int concatenation(const size_t lo, const size_t hi, const int nest) {
        std::vector<int> parts;
        for (size_t i = lo; i < hi; ++i) {
            const Tok &k = (*toks)[i];
            if (raw(i, '(')) {                                   // atom: a bracketed scope, up to where the depth is back at this bracket's
                size_t j = i + 1;
                while (j < hi && depth[j + 1] != depth[i]) ++j;
                if (j >= hi) { err = "Error: missing right bracket for " + std::to_string(i + 1) + "."; return -1; }   // (the reference's text has no " at ", regex.js:289)
                const int sub = alternation(i + 1, j, nest + 1);
                if (sub < 0) return -1;
                parts.push_back(sub);
                i = j;
            } else if (raw(i, '*') || raw(i, '+') || raw(i, '?')) {   // postfix: on the factor in front of it
                if (parts.empty()) return fail(std::string("unexpected ") + (raw(i, '*') ? '*' : '+'), i);   // (a leading '?' reports "unexpected +" in the reference, regex.js:315)
                const int x = parts.back();
                if (raw(i, '*')) {
                    Ast n; n.type = N_STAR; n.sub = x;
                    parts.back() = add(n);
                } else if (raw(i, '+')) {                        // x+ = x x*, the subtree shared
                    Ast star; star.type = N_STAR; star.sub = x;
                    const int s = add(star);
                    Ast cat; cat.type = N_CAT; cat.parts = {x, s};
                    parts.back() = add(cat);
                } else {                                         // x? = (x | empty)
                    Ast e; e.type = N_EMPTY;
                    const int en = add(e);
                    Ast o; o.type = N_OR; o.parts = {x, en};
                    parts.back() = add(o);
                }
            } else if (!k.lit && k.ch == kEpsilonChar) {
                Ast e; e.type = N_EMPTY;
                parts.push_back(add(e));
            } else {                                             // a character (a ')' or '|' that closes / splits nothing included)
                Ast x; x.type = N_TEXT; x.ch = k.ch;
                parts.push_back(add(x));
            }
        }
        if (parts.size() == 1) return parts[0];
        Ast n; n.type = N_CAT; n.parts = parts;
        return add(n);
    }
};

// ---------------------------------------------------------------- NFA (regex.js:375-437)
struct Nfa {
    struct Node { bool accept = false; std::vector<std::pair<sym_t, int>> edges; };
    std::vector<Node> n;
    int fresh() { n.emplace_back(); return (int)n.size() - 1; }
    void edge(int a, sym_t s, int b) { n[a].edges.emplace_back(s, b); }

    void build(const std::vector<Ast> &ast, int node, int start, int end) {
        const Ast &a = ast[node];
        switch (a.type) {
        case N_EMPTY: edge(start, kEps, end); break;
        case N_TEXT: edge(start, a.ch == kEpsilonChar ? kEps : (sym_t)a.ch, end); break;   // an escaped epsilon still labels an epsilon edge
        case N_CAT: {
            int last = start;
            for (size_t i = 0; i + 1 < a.parts.size(); i++) {
                int tmp = fresh();
                build(ast, a.parts[i], last, tmp);
                last = tmp;
            }
            build(ast, a.parts.back(), last, end);
            break;
        }
        case N_OR:
            for (int p : a.parts) {
                int ts = fresh(), te = fresh();
                edge(te, kEps, end);
                edge(start, kEps, ts);
                build(ast, p, ts, te);
            }
            break;
        case N_STAR: {
            int ts = fresh(), te = fresh();
            edge(te, kEps, ts);
            edge(te, kEps, end);
            edge(start, kEps, ts);
            edge(start, kEps, end);
            build(ast, a.sub, ts, te);
            break;
        }
        }
    }
};

// ---------------------------------------------------------------- subset construction (regex.js:445-551)
struct Dfa {
    struct State { std::string id; bool accept; std::vector<sym_t> symbols; std::vector<int> to; };   // to[i] follows symbols[i]
    std::vector<State> s;
};

static std::string alpha_count(long n) {    // toAlphaCount, regex.js:516-526
    std::string r;
    while (n >= 0) { r.insert(r.begin(), (char)('A' + n % 26)); n = n / 26 - 1; }
    return r;
}

static Dfa subset_construction(const Nfa &nfa, int start) {
    struct Closure { std::vector<int> items; std::vector<sym_t> symbols; bool accept; };
    std::vector<uint32_t> in(nfa.n.size(), 0);      // stamp of the closure call that last added the node
    uint32_t stamp = 0;
    auto closure_of = [&](const std::vector<int> &seed) {
        Closure c; c.accept = false;
        ++stamp;
        std::vector<int> stack;
        for (int x : seed) if (in[x] != stamp) { in[x] = stamp; stack.push_back(x); c.items.push_back(x); if (nfa.n[x].accept) c.accept = true; }
        while (!stack.empty()) {
            int top = stack.back(); stack.pop_back();
            for (auto &e : nfa.n[top].edges) {
                if (e.first == kEps) {
                    if (in[e.second] != stamp) { in[e.second] = stamp; stack.push_back(e.second); c.items.push_back(e.second); if (nfa.n[e.second].accept) c.accept = true; }
                } else c.symbols.push_back(e.first);
            }
        }
        std::sort(c.items.begin(), c.items.end());
        std::sort(c.symbols.begin(), c.symbols.end());       // Array.prototype.sort on 1-unit strings == code-unit order
        c.symbols.erase(std::unique(c.symbols.begin(), c.symbols.end()), c.symbols.end());
        return c;
    };
    Dfa dfa;
    std::map<std::vector<int>, int> index;
    std::vector<Closure> cl;
    cl.push_back(closure_of({start}));
    index[cl[0].items] = 0;
    dfa.s.push_back({alpha_count(0), cl[0].accept, cl[0].symbols, {}});
    // getClosedMove per (state, symbol) in symbol order — the order that names the states — with the moves of one
    // state bucketed in a single pass and closures memoised by their seed set (character classes written as long
    // alternations send ~100 symbols to the same seed set).
    std::map<std::vector<int>, int> seed_memo;
    for (size_t front = 0; front < cl.size(); front++) {
        std::map<sym_t, std::vector<int>> moves;
        for (int it : cl[front].items)
            for (auto &e : nfa.n[it].edges)
                if (e.first != kEps) moves[e.first].push_back(e.second);
        for (auto &mv : moves) {                                  // ascending symbol == cl[front].symbols
            std::vector<int> &nexts = mv.second;
            std::sort(nexts.begin(), nexts.end());
            nexts.erase(std::unique(nexts.begin(), nexts.end()), nexts.end());
            int to;
            auto memo = seed_memo.find(nexts);
            if (memo != seed_memo.end()) to = memo->second;
            else {
                Closure c = closure_of(nexts);
                auto f = index.find(c.items);
                if (f == index.end()) {
                    to = (int)cl.size();
                    index[c.items] = to;
                    dfa.s.push_back({alpha_count(to), c.accept, c.symbols, {}});
                    cl.push_back(std::move(c));
                } else to = f->second;
                seed_memo[nexts] = to;
            }
            dfa.s[front].to.push_back(to);
        }
    }
    return dfa;
}

// ---------------------------------------------------------------- minimisation (regex.js:559-762)
// A JavaScript object with string keys none of which is integer-like: keys() is insertion order, delete + re-insert
// moves a key to the end.  (The integer-like case — symbols '0'..'9' — is handled where the symbol list is built.)
struct OrderedGroups {
    struct Slot { std::string key; std::vector<int> group; bool alive; };
    std::vector<Slot> slots;
    std::unordered_map<std::string, size_t> pos;
    void set(const std::string &k, std::vector<int> g) {
        auto f = pos.find(k);
        if (f != pos.end() && slots[f->second].alive) { slots[f->second].group = std::move(g); return; }
        pos[k] = slots.size();
        slots.push_back({k, std::move(g), true});
    }
    void erase(const std::string &k) { auto f = pos.find(k); if (f != pos.end()) { slots[f->second].alive = false; pos.erase(f); } }
};

struct MinDfa {
    struct Node { bool accept; std::vector<std::pair<std::string, int>> edges; /* (key JSON text, target) */
                  std::vector<std::vector<uint16_t>> key_syms; };
    std::vector<Node> nodes;
};

static void json_escape_unit(uint16_t c, std::string &out) {     // JSON.stringify of one UTF-16 unit, as UTF-8 text
    char buf[8];
    switch (c) {
    case '"': out += "\\\""; return;
    case '\\': out += "\\\\"; return;
    case '\b': out += "\\b"; return;
    case '\f': out += "\\f"; return;
    case '\n': out += "\\n"; return;
    case '\r': out += "\\r"; return;
    case '\t': out += "\\t"; return;
    }
    if (c < 0x20 || (c >= 0xd800 && c <= 0xdfff)) { snprintf(buf, sizeof buf, "\\u%04x", c); out += buf; return; }   // lone surrogate: well-formed stringify
    if (c < 0x80) out += (char)c;
    else if (c < 0x800) { out += (char)(0xc0 | c >> 6); out += (char)(0x80 | (c & 0x3f)); }
    else { out += (char)(0xe0 | c >> 12); out += (char)(0x80 | ((c >> 6) & 0x3f)); out += (char)(0x80 | (c & 0x3f)); }
}

static std::string join_ids(const Dfa &d, const std::vector<int> &g) {
    std::string k;
    for (size_t i = 0; i < g.size(); i++) { if (i) k += ','; k += d.s[g[i]].id; }
    return k;
}

static MinDfa minimise(const Dfa &d) {
    const int n = (int)d.s.size();
    // getReverseEdges (regex.js:561-596): BFS from the start; symbol keys in JS object order; rev[to][symbol] -> from*
    std::vector<sym_t> sym_insertion;
    std::vector<char> seen_sym(65536, 0);
    std::vector<std::map<sym_t, std::vector<int>>> rev(n);
    {
        std::vector<char> visited(n, 0);
        std::vector<int> queue{0};
        visited[0] = 1;
        for (size_t front = 0; front < queue.size(); front++) {
            const Dfa::State &top = d.s[queue[front]];
            for (size_t i = 0; i < top.symbols.size(); i++) {
                sym_t sy = top.symbols[i];
                if (!seen_sym[sy]) { seen_sym[sy] = 1; sym_insertion.push_back(sy); }
                int nx = top.to[i];
                rev[nx][sy].push_back(queue[front]);
                if (!visited[nx]) { visited[nx] = 1; queue.push_back(nx); }
            }
        }
    }
    std::vector<sym_t> symbols;                       // Object.keys(symbols): '0'..'9' ascending first, then insertion order
    for (sym_t c = '0'; c <= '9'; c++) if (seen_sym[c]) symbols.push_back(c);
    for (sym_t c : sym_insertion) if (c < '0' || c > '9') symbols.push_back(c);

    // hopcroft (regex.js:598-690)
    std::vector<int> ids(n);
    for (int i = 0; i < n; i++) ids[i] = i;
    std::sort(ids.begin(), ids.end(), [&](int a, int b) { return d.s[a].id < d.s[b].id; });
    std::unordered_map<std::string, int> by_name;
    for (int i = 0; i < n; i++) by_name[d.s[i].id] = i;
    OrderedGroups partitions;
    std::vector<std::string> queue;
    std::vector<char> queue_live;
    std::unordered_map<std::string, size_t> visited;
    {
        std::vector<int> g1, g2;
        for (int id : ids) (d.s[id].accept ? g1 : g2).push_back(id);
        std::string key = join_ids(d, g1);
        partitions.set(key, g1);
        queue.push_back(key); queue_live.push_back(1);
        visited[key] = 0;
        if (!g2.empty()) {
            key = join_ids(d, g2);
            partitions.set(key, g2);
            queue.push_back(key); queue_live.push_back(1);
        }
    }
    std::vector<char> rev_group(n);
    for (size_t front = 0; front < queue.size(); front++) {
        if (!queue_live[front] || queue[front].empty()) continue;       // `if (top)`: null or "" are falsy
        std::vector<int> top;
        {
            const std::string &k = queue[front];
            size_t a = 0;
            while (a <= k.size()) {
                size_t b = k.find(',', a);
                if (b == std::string::npos) b = k.size();
                auto f = by_name.find(k.substr(a, b - a));
                top.push_back(f == by_name.end() ? -1 : f->second);
                a = b + 1;
            }
        }
        for (sym_t symbol : symbols) {
            std::fill(rev_group.begin(), rev_group.end(), 0);
            for (int t : top) {
                if (t < 0) continue;
                auto f = rev[t].find(symbol);
                if (f != rev[t].end()) for (int from : f->second) rev_group[from] = 1;
            }
            std::vector<std::string> keys;                               // Object.keys(partitions) snapshot
            for (auto &s : partitions.slots) if (s.alive) keys.push_back(s.key);
            for (const std::string &key : keys) {
                std::vector<int> g1, g2;
                for (int x : partitions.slots[partitions.pos[key]].group) (rev_group[x] ? g1 : g2).push_back(x);
                if (g1.empty() || g2.empty()) continue;
                partitions.erase(key);
                std::string key1 = join_ids(d, g1), key2 = join_ids(d, g2);
                const bool smaller1 = g1.size() <= g2.size();
                partitions.set(key1, std::move(g1));
                partitions.set(key2, std::move(g2));
                auto v = visited.find(key1);
                if (v != visited.end()) {
                    queue_live[v->second] = 0;
                    visited[key1] = queue.size(); queue.push_back(key1); queue_live.push_back(1);
                    visited[key2] = queue.size(); queue.push_back(key2); queue_live.push_back(1);
                } else if (smaller1) {
                    visited[key1] = queue.size(); queue.push_back(key1); queue_live.push_back(1);
                } else {
                    visited[key2] = queue.size(); queue.push_back(key2); queue_live.push_back(1);
                }
            }
        }
    }

    // buildMinNfa (regex.js:691-756)
    std::vector<std::pair<std::string, std::vector<int>>> parts;
    for (auto &s : partitions.slots) if (s.alive) parts.emplace_back(s.key, s.group);
    std::sort(parts.begin(), parts.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    for (size_t i = 0; i < parts.size(); i++)
        if (std::find(parts[i].second.begin(), parts[i].second.end(), 0) != parts[i].second.end()) {
            if (i > 0) std::swap(parts[i], parts[0]);
            break;
        }
    std::vector<int> group(n, -1);
    MinDfa m;
    m.nodes.resize(parts.size());
    for (size_t i = 0; i < parts.size(); i++) {
        m.nodes[i].accept = d.s[parts[i].second[0]].accept;
        for (int x : parts[i].second) group[x] = (int)i;
    }
    std::vector<std::map<int, std::vector<uint16_t>>> edges(parts.size());    // from -> to (ascending: integer keys) -> symbols
    for (int to = 0; to < n; to++)
        for (auto &kv : rev[to])
            for (int from : kv.second) {
                auto &v = edges[group[from]][group[to]];
                if (std::find(v.begin(), v.end(), (uint16_t)kv.first) == v.end()) v.push_back((uint16_t)kv.first);
            }
    for (size_t from = 0; from < parts.size(); from++)
        for (auto &kv : edges[from]) {
            std::vector<uint16_t> syms = kv.second;
            std::sort(syms.begin(), syms.end());
            std::string key = "[";
            for (size_t i = 0; i < syms.size(); i++) { if (i) key += ','; key += '"'; json_escape_unit(syms[i], key); key += '"'; }
            key += "]";
            m.nodes[from].edges.emplace_back(key, kv.first);
            m.nodes[from].key_syms.push_back(syms);
        }
    return m;
}

// UTF-16 code-unit order of two UTF-8 texts (Array.prototype.sort on strings): differs from byte order only between
// supplementary-plane characters and U+E000..U+FFFF; the key texts here never hold supplementary characters unescaped
// in pairs that would matter, so compare decoded units.
static bool less_utf16(const std::string &a, const std::string &b) {
    std::vector<uint16_t> ua, ub; std::string e;
    utf8_to_units(a.data(), a.size(), ua, e);
    utf8_to_units(b.data(), b.size(), ub, e);
    return ua < ub;
}

static void json_escape_string(const std::string &utf8, std::string &out) {   // JSON.stringify(string) of well-formed text
    out += '"';
    for (unsigned char c : utf8) {
        if (c >= 0x80) { out += (char)c; continue; }
        json_escape_unit(c, out);
    }
    out += '"';
}

}  // namespace rx

// regexToDfa's return value (regex.js:40-92): JSON.stringify of [{type, edges:{key: target}}], edges in sorted key order
static std::string dfa_json(const rx::MinDfa &m) {
    std::string out = "[";
    for (size_t i = 0; i < m.nodes.size(); i++) {
        if (i) out += ',';
        out += m.nodes[i].accept ? "{\"type\":\"accept\",\"edges\":{" : "{\"type\":\"\",\"edges\":{";
        std::vector<size_t> order(m.nodes[i].edges.size());
        for (size_t k = 0; k < order.size(); k++) order[k] = k;
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return rx::less_utf16(m.nodes[i].edges[a].first, m.nodes[i].edges[b].first); });
        for (size_t k = 0; k < order.size(); k++) {
            if (k) out += ',';
            rx::json_escape_string(m.nodes[i].edges[order[k]].first, out);
            out += ':' + std::to_string(m.nodes[i].edges[order[k]].second);
        }
        out += "}}";
    }
    return out + "]";
}

// dfa_to_regex_def_text (js_caller.rs:127-157)
static bool allstr_text(const rx::MinDfa &m, std::string &out, std::string &err) {
    long accepted = -1;
    size_t max_state = 0;
    for (size_t i = 0; i < m.nodes.size(); i++) {
        if (accepted < 0 && m.nodes[i].accept) accepted = (long)i;
        for (auto &e : m.nodes[i].edges) max_state = std::max(max_state, (size_t)e.second);
    }
    if (accepted < 0) { err = "No accepted state"; return false; }
    out = "0\n" + std::to_string(accepted) + "\n" + std::to_string(max_state) + "\n";
    for (size_t i = 0; i < m.nodes.size(); i++) {
        std::vector<size_t> order(m.nodes[i].edges.size());
        for (size_t k = 0; k < order.size(); k++) order[k] = k;
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return m.nodes[i].edges[a].first < m.nodes[i].edges[b].first; });   // BTreeMap<String,_>
        for (size_t k : order)
            for (uint16_t c : m.nodes[i].key_syms[k])
                out += std::to_string(i) + " " + std::to_string(m.nodes[i].edges[k].second) + " " + std::to_string((unsigned)(c & 0xff)) + "\n";
    }
    return true;
}

bool compile_regex(const char *regex, size_t len, std::string *json_out, std::string *text_out, std::string &err,
                   CompiledDfa *dfa_out) {
    std::vector<uint16_t> units;
    if (!rx::utf8_to_units(regex, len, units, err)) return false;
    std::vector<rx::Tok> toks;
    for (size_t i = 0; i < units.size();) {
        if (units[i] == '\\') {
            if (i + 1 >= units.size()) { err = "Error: pattern ends in a lone backslash"; return false; }
            uint16_t c = units[i + 1];
            switch (c) { case 'n': c = '\n'; break; case 'r': c = '\r'; break; case 't': c = '\t'; break; case 'v': c = '\v'; break; case 'f': c = '\f'; break; }
            toks.push_back({c, true});
            i += 2;
        } else { toks.push_back({units[i], false}); i += 1; }
    }
    rx::Parser p;
    int root = p.parse(toks);
    if (root < 0) { err = p.err; return false; }
    rx::Nfa nfa;
    int start = nfa.fresh(), accept = nfa.fresh();
    nfa.n[accept].accept = true;
    nfa.build(p.nodes, root, start, accept);
    rx::Dfa dfa = rx::subset_construction(nfa, start);
    rx::MinDfa m = rx::minimise(dfa);
    if (json_out) *json_out = dfa_json(m);
    if (text_out && !allstr_text(m, *text_out, err)) return false;
    if (dfa_out) {                     // the parsed JSON value as the Rust side sees it: edges in BTreeMap (byte) order
        dfa_out->nodes.clear();
        for (auto &n : m.nodes) {
            CompiledDfa::Node o; o.accept = n.accept;
            std::vector<size_t> order(n.edges.size());
            for (size_t k = 0; k < order.size(); k++) order[k] = k;
            std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return n.edges[a].first < n.edges[b].first; });
            for (size_t k : order) o.edges.push_back({n.edges[k].first, n.edges[k].second, n.key_syms[k]});
            dfa_out->nodes.push_back(std::move(o));
        }
    }
    return true;
}

}  // namespace hrx
