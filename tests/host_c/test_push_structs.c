/* The call sequence of bindings/rust/hrx.rs HrxHandle::new, replayed in C: a Rust caller holds AllstrRegexDef { state_lookup: HashMap<(u8, u64), (usize, u64)>, .. }
 * (src/defs.rs:26-36) and SubstrRegexDef (defs.rs:115-132) and hands them over with hrx_defs_push_allstr / hrx_defs_push_substr — the map's entries in the map's
 * own (arbitrary) iteration order, each with the line index the parser stored (defs.rs:100), which is what RegexTableConfig::load sorts by (table.rs:103-108).
 *
 * For every definition: config A = the text parsers (hrx_defs_push_*_text), config B = this file's own restatement of read_from_reader
 * (defs.rs:75-110, 209-265) -> entries SHUFFLED -> the struct entry points.  A duplicate (char, state) line is appended to the allstr text first: the
 * HashMap keeps the LAST insert (its line index and its next state), so B's deduplicated map must equal A's parse.  A and B must agree on every fixed-table
 * row (table.rs:103-198) and on every witness row of a few strings (lib.rs:316-318, 387-764).
 *
 *   test_push_structs <dir with the DFA fixtures> [gpu]      ("gpu": the witness rows come from device 0 instead of the host walk) */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/hrx.h"

#define CHECK(x) do { if ((x) != 0) { fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #x, hrx_last_error()); exit(1); } } while (0)

static char *slurp(const char *path, size_t *len, size_t extra) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(1); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *p = malloc((size_t)n + extra + 1);
    if (fread(p, 1, (size_t)n, f) != (size_t)n) { perror(path); exit(1); }
    fclose(f);
    p[n] = 0;
    *len = (size_t)n;
    return p;
}

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t rng(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

typedef struct { uint64_t cur, next, idx; uint8_t chr; } entry_t;

/* AllstrRegexDef::read_from_reader (defs.rs:75-110) as a HashMap would leave it: one entry per (char, state) key, the LAST line's value */
static size_t parse_allstr(const char *text, uint64_t head[3], entry_t **out) {
    size_t cap = 1024, n = 0, line = 0;
    entry_t *e = malloc(cap * sizeof *e);
    const char *p = text;
    while (*p) {
        const char *eol = strchr(p, '\n');
        if (!eol) eol = p + strlen(p);
        unsigned long long a = 0, b = 0, c = 0;
        const int got = sscanf(p, "%llu %llu %llu", &a, &b, &c);
        if (line < 3) {
            head[line] = a;
        } else if (got == 3) {
            const uint8_t ch = (uint8_t)c;   /* `as u8` */
            size_t k = 0;
            for (; k < n; ++k)
                if (e[k].chr == ch && e[k].cur == a) break;
            if (k == n) { if (n == cap) { cap *= 2; e = realloc(e, cap * sizeof *e); } ++n; }
            e[k].cur = a; e[k].next = b; e[k].chr = ch; e[k].idx = line;   /* insert(): the value is replaced, the key stays */
        }
        ++line;
        p = *eol ? eol + 1 : eol;
    }
    *out = e;
    return n;
}

typedef struct { size_t n_pairs, n_start, n_end; uint64_t *pc, *pn, *st, *en; } substr_t;
static size_t parse_list(const char *p, const char *eol, uint64_t **out) {
    size_t n = 0, cap = 16;
    uint64_t *v = malloc(cap * sizeof *v);
    while (p < eol) {
        while (p < eol && (*p == ' ' || *p == '\r' || *p == '\t')) ++p;
        if (p >= eol) break;
        char *end;
        const unsigned long long x = strtoull(p, &end, 10);
        if (end == p) break;
        if (n == cap) { cap *= 2; v = realloc(v, cap * sizeof *v); }
        v[n++] = x;
        p = end;
    }
    *out = v;
    return n;
}
/* SubstrRegexDef::read_from_reader (defs.rs:209-265): lines 0-2 max_length / min_position / max_position, 3 start states, 4 end states, then "cur next" */
static substr_t parse_substr(const char *text) {
    substr_t s;
    memset(&s, 0, sizeof s);
    size_t cap = 64, line = 0;
    s.pc = malloc(cap * sizeof(uint64_t)); s.pn = malloc(cap * sizeof(uint64_t));
    const char *p = text;
    while (*p) {
        const char *eol = strchr(p, '\n');
        if (!eol) eol = p + strlen(p);
        if (line == 3) s.n_start = parse_list(p, eol, &s.st);
        else if (line == 4) s.n_end = parse_list(p, eol, &s.en);
        else if (line > 4) {
            unsigned long long a, b;
            if (sscanf(p, "%llu %llu", &a, &b) == 2) {
                if (s.n_pairs == cap) { cap *= 2; s.pc = realloc(s.pc, cap * sizeof(uint64_t)); s.pn = realloc(s.pn, cap * sizeof(uint64_t)); }
                s.pc[s.n_pairs] = a; s.pn[s.n_pairs] = b; ++s.n_pairs;
            }
        }
        ++line;
        p = *eol ? eol + 1 : eol;
    }
    return s;
}

int main(int argc, char **argv) {
    const char *dir = argc > 1 ? argv[1] : "tests/golden/dfa";
    const int gpu = argc > 2 && !strcmp(argv[2], "gpu");
    const char *allstr_files[] = {"regex1_test_lookup.txt", "regex2_test_lookup.txt"};
    const char *substr_files[] = {"substr1_test_lookup.txt", "substr2_test_lookup.txt"};
    hrx_defs *A = NULL, *B = NULL;
    CHECK(hrx_defs_create(&A));
    CHECK(hrx_defs_create(&B));
    size_t total_entries = 0;
    for (int d = 0; d < 2; ++d) {
        char path[1024];
        size_t len = 0;
        snprintf(path, sizeof path, "%s/%s", dir, allstr_files[d]);
        char *text = slurp(path, &len, 0);
        /* a duplicate key (defs.rs:100): the key of the file's first transition line once more IN FRONT of it, with another next state — the map keeps the
         * later (real) line: its next state and its line index; and the same real line again at the very end: now THAT line index counts */
        unsigned long long c0, n0, ch0;
        char *first = text;
        for (int i = 0; i < 3; ++i) first = strchr(first, '\n') + 1;
        sscanf(first, "%llu %llu %llu", &c0, &n0, &ch0);
        char bogus[64];
        const size_t bl = (size_t)sprintf(bogus, "%llu %llu %llu\n", c0, n0 + 1, ch0);
        text = realloc(text, len + 2 * bl + 8);
        first = text;
        for (int i = 0; i < 3; ++i) first = strchr(first, '\n') + 1;
        memmove(first + bl, first, len - (size_t)(first - text) + 1);
        memcpy(first, bogus, bl);
        len += bl;
        if (text[len - 1] != '\n') text[len++] = '\n';
        len += (size_t)sprintf(text + len, "%llu %llu %llu\n", c0, n0, ch0);
        CHECK(hrx_defs_push_allstr_text(A, text, len));
        uint64_t head[3];
        entry_t *e = NULL;
        const size_t n = parse_allstr(text, head, &e);
        total_entries += n;
        for (size_t i = n - 1; i > 0; --i) {   /* a HashMap's iteration order is arbitrary: shuffle */
            const size_t j = (size_t)(rng() % (i + 1));
            const entry_t t = e[i]; e[i] = e[j]; e[j] = t;
        }
        uint64_t *cur = malloc(n * 8), *next = malloc(n * 8), *idx = malloc(n * 8);
        uint8_t *chr = malloc(n);
        for (size_t i = 0; i < n; ++i) { cur[i] = e[i].cur; next[i] = e[i].next; chr[i] = e[i].chr; idx[i] = e[i].idx; }
        CHECK(hrx_defs_push_allstr(B, head[0], head[1], head[2], n, cur, next, chr, idx));
        snprintf(path, sizeof path, "%s/%s", dir, substr_files[d]);
        char *stext = slurp(path, &len, 0);
        CHECK(hrx_defs_push_substr_text(A, stext, len));
        substr_t s = parse_substr(stext);
        for (size_t i = s.n_pairs - 1; i > 0; --i) {   /* HashSet order */
            const size_t j = (size_t)(rng() % (i + 1));
            uint64_t t = s.pc[i]; s.pc[i] = s.pc[j]; s.pc[j] = t;
            t = s.pn[i]; s.pn[i] = s.pn[j]; s.pn[j] = t;
        }
        CHECK(hrx_defs_push_substr(B, s.n_pairs, s.pc, s.pn, s.n_start, s.st, s.n_end, s.en));
    }
    CHECK(hrx_defs_finalize(A));
    CHECK(hrx_defs_finalize(B));
    if (total_entries != 2842 + 1274) { fprintf(stderr, "parsed %zu entries, expected 2842 + 1274\n", total_entries); return 1; }

    /* fixed tables (table.rs:103-198): transitions in line-index order, then the dummy row; endpoints */
    for (size_t d = 0; d < 2; ++d) {
        if (hrx_defs_num_transitions(A, d) != hrx_defs_num_transitions(B, d)) { fprintf(stderr, "def %zu: transition counts differ\n", d); return 1; }
        const size_t cap = hrx_defs_num_transitions(A, d) + 8;
        uint64_t *ra = calloc(cap * 4, 8), *rb = calloc(cap * 4, 8);
        const size_t na = hrx_table_transition_rows(A, d, ra, cap), nb = hrx_table_transition_rows(B, d, rb, cap);
        if (na != nb || na == 0 || na > cap || memcmp(ra, rb, na * 32)) { fprintf(stderr, "def %zu: transition rows differ (%zu vs %zu)\n", d, na, nb); return 1; }
        const size_t ecap = 4096;
        uint64_t *ea = calloc(ecap * 3, 8), *eb = calloc(ecap * 3, 8);
        const size_t ma = hrx_table_endpoint_rows(A, d, ea, ecap), mb = hrx_table_endpoint_rows(B, d, eb, ecap);
        if (ma != mb || ma > ecap || memcmp(ea, eb, ma * 24)) { fprintf(stderr, "def %zu: endpoint rows differ (%zu vs %zu)\n", d, ma, mb); return 1; }
        free(ra); free(rb); free(ea); free(eb);
    }

    /* witness rows of the reference's own test strings (lib.rs:1073, 1100, 1323) + one that takes the duplicate's transition */
    const char *strings[] = {"email was meant for @y. Also for x.", "email was meant for @y. Also for x", "email was meant for @yz", "", "e"};
    enum { NS = 5, M = 64, STRIDE = 64, D = 2 };
    uint8_t chars[NS * STRIDE];
    uint32_t lens[NS];
    memset(chars, 0xaa, sizeof chars);
    for (int i = 0; i < NS; ++i) { lens[i] = (uint32_t)strlen(strings[i]); memcpy(chars + i * STRIDE, strings[i], lens[i]); }
    uint32_t rec[2][NS * M * D];
    uint16_t msk[2][NS * M];
    uint64_t st[2][NS];
    hrx_defs *both[2] = {A, B};
    for (int k = 0; k < 2; ++k) {
        hrx_ctx *ctx = NULL;
        CHECK(hrx_ctx_create(both[k], gpu ? 0 : HRX_DEVICE_NONE, &ctx));
        if (gpu) CHECK(hrx_ctx_set_host_threshold(ctx, 0));   /* never the host walk: the rows come from the device */
        CHECK(hrx_witness_batch_host(ctx, chars, STRIDE, lens, NS, M, rec[k], msk[k], st[k]));
        hrx_ctx_destroy(ctx);
    }
    if (memcmp(rec[0], rec[1], sizeof rec[0]) || memcmp(msk[0], msk[1], sizeof msk[0]) || memcmp(st[0], st[1], sizeof st[0])) { fprintf(stderr, "witness rows differ\n"); return 1; }
    /* sanity: the first string is the reference's accepted case — both defs end in their accept states (status = ok | accept mask 3 << 8) */
    if (st[0][0] != (3ull << 8)) { fprintf(stderr, "status of the accepted string: %llx\n", (unsigned long long)st[0][0]); return 1; }
    hrx_defs_destroy(A);
    hrx_defs_destroy(B);
    printf(gpu ? "gpu ok\n" : "host ok\n");
    return 0;
}
