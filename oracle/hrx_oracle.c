/*
 * hrx_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's witness-generation path
 * (zkemail/halo2-regex @ /root/reference).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / the timed CPU baseline — the product path (halo2_regex_amd/csrc)
 * never links, loads or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against
 * every known-answer the reference's own tests hold for the path
 * (tests/golden/reference_tests.json, extracted from src/lib.rs:1067-1470 and
 * examples/regex.rs:185-199) on the reference's own DFA fixtures
 * (tests/golden/dfa/, copied data files).  The Rust itself cannot be built in
 * this image (no cargo/rustc; un-vendored git dependencies), so there is no
 * oracle/_ref build.
 *
 * It deliberately keeps the reference's data structures and loop order:
 *   - AllstrRegexDef.state_lookup : hash map (u8 char, u64 state) -> (line idx, next)   src/defs.rs:28
 *   - SubstrRegexDef.valid_state_transitions : hash set of (u64,u64)                  src/defs.rs:127
 *   - start_states / end_states : vectors scanned linearly (Vec::contains)            src/defs.rs:129-131
 * so that it can also serve as the "port" CPU baseline (same asymptotics and
 * memory behaviour as src/lib.rs:804-888).
 *
 * Each function cites the reference lines it restates.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <pthread.h>

/* ------------------------------------------------------------------ */
/* containers                                                          */
/* ------------------------------------------------------------------ */

typedef struct {
    uint8_t  used;
    uint8_t  ch;
    uint64_t state;
    uint64_t line_idx; /* "index of the state transitions" = text line index, defs.rs:84,100 */
    uint64_t next;
} lookup_slot;

typedef struct {
    lookup_slot *slots;
    size_t cap; /* power of two */
    size_t len;
} lookup_map;

typedef struct {
    uint8_t  used;
    uint64_t a, b;
} pair_slot;

typedef struct {
    pair_slot *slots;
    size_t cap;
    size_t len;
} pair_set;

typedef struct {
    uint64_t *v;
    size_t len, cap;
} u64_vec;

static uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

static void map_init(lookup_map *m) { m->cap = 64; m->len = 0; m->slots = calloc(m->cap, sizeof(lookup_slot)); }
static void map_free(lookup_map *m) { free(m->slots); m->slots = NULL; }
static lookup_slot *map_find(const lookup_map *m, uint8_t ch, uint64_t state) {
    size_t i = (size_t)mix64(state * 257u + ch) & (m->cap - 1);
    for (;;) {
        lookup_slot *s = &m->slots[i];
        if (!s->used) return NULL;
        if (s->ch == ch && s->state == state) return s;
        i = (i + 1) & (m->cap - 1);
    }
}
static void map_insert(lookup_map *m, uint8_t ch, uint64_t state, uint64_t line_idx, uint64_t next);
static void map_grow(lookup_map *m) {
    lookup_map n; n.cap = m->cap * 2; n.len = 0; n.slots = calloc(n.cap, sizeof(lookup_slot));
    for (size_t i = 0; i < m->cap; i++)
        if (m->slots[i].used) map_insert(&n, m->slots[i].ch, m->slots[i].state, m->slots[i].line_idx, m->slots[i].next);
    free(m->slots); *m = n;
}
/* HashMap::insert: a later duplicate key overwrites the value (defs.rs:100). */
static void map_insert(lookup_map *m, uint8_t ch, uint64_t state, uint64_t line_idx, uint64_t next) {
    lookup_slot *f = map_find(m, ch, state);
    if (f) { f->line_idx = line_idx; f->next = next; return; }
    if ((m->len + 1) * 2 > m->cap) map_grow(m);
    size_t i = (size_t)mix64(state * 257u + ch) & (m->cap - 1);
    while (m->slots[i].used) i = (i + 1) & (m->cap - 1);
    m->slots[i].used = 1; m->slots[i].ch = ch; m->slots[i].state = state;
    m->slots[i].line_idx = line_idx; m->slots[i].next = next; m->len++;
}

static void set_init(pair_set *s) { s->cap = 16; s->len = 0; s->slots = calloc(s->cap, sizeof(pair_slot)); }
static void set_free(pair_set *s) { free(s->slots); s->slots = NULL; }
static int set_contains(const pair_set *s, uint64_t a, uint64_t b) {
    size_t i = (size_t)mix64(a * 0x9e3779b97f4a7c15ULL + b) & (s->cap - 1);
    for (;;) {
        const pair_slot *p = &s->slots[i];
        if (!p->used) return 0;
        if (p->a == a && p->b == b) return 1;
        i = (i + 1) & (s->cap - 1);
    }
}
static void set_insert(pair_set *s, uint64_t a, uint64_t b) {
    if (set_contains(s, a, b)) return;
    if ((s->len + 1) * 2 > s->cap) {
        pair_set n; n.cap = s->cap * 2; n.len = 0; n.slots = calloc(n.cap, sizeof(pair_slot));
        for (size_t i = 0; i < s->cap; i++) if (s->slots[i].used) set_insert(&n, s->slots[i].a, s->slots[i].b);
        free(s->slots); *s = n;
    }
    size_t i = (size_t)mix64(a * 0x9e3779b97f4a7c15ULL + b) & (s->cap - 1);
    while (s->slots[i].used) i = (i + 1) & (s->cap - 1);
    s->slots[i].used = 1; s->slots[i].a = a; s->slots[i].b = b; s->len++;
}

static void vec_push(u64_vec *v, uint64_t x) {
    if (v->len == v->cap) { v->cap = v->cap ? v->cap * 2 : 8; v->v = realloc(v->v, v->cap * sizeof(uint64_t)); }
    v->v[v->len++] = x;
}
static int vec_contains(const u64_vec *v, uint64_t x) { /* Vec::contains, lib.rs:866,879 */
    for (size_t i = 0; i < v->len; i++) if (v->v[i] == x) return 1;
    return 0;
}

/* ------------------------------------------------------------------ */
/* data model (src/defs.rs:17-36, 115-132)                             */
/* ------------------------------------------------------------------ */

typedef struct {
    uint64_t max_length, min_position, max_position; /* unused by the chip, defs.rs:118-125 */
    pair_set valid_state_transitions;
    u64_vec  start_states, end_states;
} substr_def;

typedef struct {
    lookup_map state_lookup;
    uint64_t first_state_val, accepted_state_val, largest_state_val;
    substr_def *substrs;
    size_t n_substrs;
} regex_defs;

typedef struct orc {
    regex_defs *defs;
    size_t n_defs;
} orc;

orc *orc_new(void) { return calloc(1, sizeof(orc)); }

void orc_free(orc *o) {
    if (!o) return;
    for (size_t d = 0; d < o->n_defs; d++) {
        map_free(&o->defs[d].state_lookup);
        for (size_t j = 0; j < o->defs[d].n_substrs; j++) {
            set_free(&o->defs[d].substrs[j].valid_state_transitions);
            free(o->defs[d].substrs[j].start_states.v);
            free(o->defs[d].substrs[j].end_states.v);
        }
        free(o->defs[d].substrs);
    }
    free(o->defs); free(o);
}

size_t orc_num_defs(const orc *o) { return o->n_defs; }
size_t orc_num_substrs(const orc *o, size_t d) { return o->defs[d].n_substrs; }
uint64_t orc_first_state(const orc *o, size_t d) { return o->defs[d].first_state_val; }
uint64_t orc_accepted_state(const orc *o, size_t d) { return o->defs[d].accepted_state_val; }
uint64_t orc_largest_state(const orc *o, size_t d) { return o->defs[d].largest_state_val; }
size_t orc_num_transitions(const orc *o, size_t d) { return o->defs[d].state_lookup.len; }

/* One text line -> Vec<u64> exactly like
 *   line.split_whitespace().map(|s| s.parse::<u64>().expect(..))        defs.rs:86-92, 220-226
 * Returns the number of elements, or -1 on a parse failure (the reference panics). */
static long parse_line(const char *p, const char *end, uint64_t *out, size_t cap) {
    long n = 0;
    while (p < end) {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\v' || *p == '\f')) p++;
        if (p >= end) break;
        const char *tok = p;
        if (*p == '+') p++; /* u64::from_str accepts a leading '+' */
        if (p >= end || *p < '0' || *p > '9') return -1;
        uint64_t v = 0;
        while (p < end && *p >= '0' && *p <= '9') {
            uint64_t d = (uint64_t)(*p - '0');
            if (v > (UINT64_MAX - d) / 10) return -1; /* overflow -> parse error */
            v = v * 10 + d; p++;
        }
        if (p < end && !(*p == ' ' || *p == '\t' || *p == '\r' || *p == '\v' || *p == '\f')) return -1;
        (void)tok;
        if ((size_t)n < cap) out[n] = v;
        n++;
    }
    return n;
}

/* AllstrRegexDef::read_from_reader — src/defs.rs:75-110.
 * Pushes a new RegexDefs { allstr, substrs: vec![] } (defs.rs:17-22).
 * Returns 0, or -(line_idx+1) where the reference would panic. */
int orc_push_allstr_text(orc *o, const char *txt, size_t len) {
    regex_defs rd; memset(&rd, 0, sizeof rd);
    map_init(&rd.state_lookup);
    const char *p = txt, *end = txt + len;
    uint64_t idx = 0;
    /* BufRead::lines(): split on '\n'; a trailing '\n' does not yield an extra empty line */
    while (p < end) {
        const char *nl = memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        uint64_t el[8];
        long n = parse_line(p, le, el, 8);
        if (n < 0) { map_free(&rd.state_lookup); return -(int)(idx + 1); }
        if (idx <= 2) {
            if (n < 1) { map_free(&rd.state_lookup); return -(int)(idx + 1); } /* elements[0] out of bounds */
            if (idx == 0) rd.first_state_val = el[0];            /* defs.rs:93-94 */
            else if (idx == 1) rd.accepted_state_val = el[0];    /* defs.rs:95-96 */
            else rd.largest_state_val = el[0];                   /* defs.rs:97-98 */
        } else {
            if (n < 3) { map_free(&rd.state_lookup); return -(int)(idx + 1); } /* elements[2] out of bounds */
            /* state_lookup.insert((elements[2] as u8, elements[0]), (idx, elements[1]))   defs.rs:100 */
            map_insert(&rd.state_lookup, (uint8_t)el[2], el[0], idx, el[1]);
        }
        idx++;
        p = nl ? nl + 1 : end;
    }
    o->defs = realloc(o->defs, (o->n_defs + 1) * sizeof(regex_defs));
    o->defs[o->n_defs++] = rd;
    return 0;
}

/* SubstrRegexDef::read_from_reader — src/defs.rs:209-265.  Appends to the last RegexDefs.substrs. */
int orc_push_substr_text(orc *o, const char *txt, size_t len) {
    if (o->n_defs == 0) return -1000000;
    substr_def sd; memset(&sd, 0, sizeof sd);
    set_init(&sd.valid_state_transitions);
    const char *p = txt, *end = txt + len;
    uint64_t idx = 0;
    int rc = 0;
    size_t cap = len / 2 + 8;
    uint64_t *el = malloc(cap * sizeof(uint64_t));
    while (p < end) {
        const char *nl = memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        long n = parse_line(p, le, el, cap);
        if (n < 0) { rc = -(int)(idx + 1); break; }
        if (idx <= 2) {
            if (n < 1) { rc = -(int)(idx + 1); break; }
            if (idx == 0) sd.max_length = el[0];        /* defs.rs:227-228 */
            else if (idx == 1) sd.min_position = el[0]; /* defs.rs:229-230 */
            else sd.max_position = el[0];               /* defs.rs:231-232 */
        } else if (idx == 3) {
            for (long i = 0; i < n; i++) vec_push(&sd.start_states, el[i]); /* defs.rs:233-234 */
        } else if (idx == 4) {
            for (long i = 0; i < n; i++) vec_push(&sd.end_states, el[i]);   /* defs.rs:235-236 */
        } else {
            if (n < 2) { rc = -(int)(idx + 1); break; }
            set_insert(&sd.valid_state_transitions, el[0], el[1]);          /* defs.rs:238 */
        }
        idx++;
        p = nl ? nl + 1 : end;
    }
    free(el);
    if (rc) { set_free(&sd.valid_state_transitions); free(sd.start_states.v); free(sd.end_states.v); return rc; }
    regex_defs *rd = &o->defs[o->n_defs - 1];
    rd->substrs = realloc(rd->substrs, (rd->n_substrs + 1) * sizeof(substr_def));
    rd->substrs[rd->n_substrs++] = sd;
    return 0;
}

/* ------------------------------------------------------------------ */
/* derive_* — src/lib.rs:804-888                                       */
/* ------------------------------------------------------------------ */

/* derive_states — src/lib.rs:804-823.
 * states: D x (n+1), row-major.  Returns 0, or 1 where the reference panics with
 * "The transition from {state} by {char} is invalid!" (lib.rs:817); then
 * panic_info = {def idx, position, state, char}.  Defs are walked in order, so the
 * first panic is the lowest def, then the lowest position. */
int orc_derive_states(const orc *o, const uint8_t *chars, size_t n, uint64_t *states, uint64_t *panic_info) {
    for (size_t d = 0; d < o->n_defs; d++) {
        const regex_defs *defs = &o->defs[d];
        uint64_t *s = states + d * (n + 1);
        s[0] = defs->first_state_val;                                       /* lib.rs:807 */
        for (size_t c = 0; c < n; c++) {
            const lookup_slot *nx = map_find(&defs->state_lookup, chars[c], s[c]); /* lib.rs:810 */
            if (!nx) {
                if (panic_info) { panic_info[0] = d; panic_info[1] = c; panic_info[2] = s[c]; panic_info[3] = chars[c]; }
                return 1;                                                   /* lib.rs:817 */
            }
            s[c + 1] = nx->next;                                            /* lib.rs:816 */
        }
    }
    return 0;
}

/* derive_substr_ids — src/lib.rs:825-845.  sids: D x n. */
void orc_derive_substr_ids(const orc *o, const uint64_t *states, size_t n, uint64_t *sids) {
    uint64_t substr_id_offset = 1;                                          /* lib.rs:827 */
    for (size_t d = 0; d < o->n_defs; d++) {
        const regex_defs *defs = &o->defs[d];
        const uint64_t *s = states + d * (n + 1);
        uint64_t *id = sids + d * n;
        for (size_t i = 0; i < n; i++) {
            id[i] = 0;
            for (size_t j = 0; j < defs->n_substrs; j++) {                  /* first match wins, lib.rs:831-840 */
                if (set_contains(&defs->substrs[j].valid_state_transitions, s[i], s[i + 1])) {
                    id[i] = substr_id_offset + j;
                    break;
                }
            }
        }
        substr_id_offset += defs->n_substrs;                                /* lib.rs:842 */
    }
}

/* derive_is_start_end — src/lib.rs:847-888.  is_start, is_end: D x (n+1) bytes. */
void orc_derive_is_start_end(const orc *o, const uint64_t *states, const uint64_t *sids, size_t n,
                             uint8_t *is_start, uint8_t *is_end) {
    uint64_t substr_id_offset = 1;                                          /* lib.rs:854 */
    for (size_t d = 0; d < o->n_defs; d++) {
        const regex_defs *defs = &o->defs[d];
        const uint64_t *s = states + d * (n + 1);
        const uint64_t *id = sids + d * n;
        uint8_t *st = is_start + d * (n + 1);
        uint8_t *en = is_end + d * (n + 1);
        for (size_t i = 0; i < n; i++) {                                    /* lib.rs:857-868 */
            if (id[i] == 0) { st[i] = 0; continue; }
            st[i] = (uint8_t)vec_contains(&defs->substrs[id[i] - substr_id_offset].start_states, s[i]);
        }
        st[n] = 0;                                                          /* lib.rs:869 */
        en[0] = 0;                                                          /* lib.rs:882 */
        for (size_t i = 0; i < n; i++) {                                    /* lib.rs:870-881 */
            if (id[i] == 0) { en[i + 1] = 0; continue; }
            en[i + 1] = (uint8_t)vec_contains(&defs->substrs[id[i] - substr_id_offset].end_states, s[i + 1]);
        }
        substr_id_offset += defs->n_substrs;                                /* lib.rs:885 */
    }
}

/* ------------------------------------------------------------------ */
/* match_substrs, integer content — src/lib.rs:311-773                 */
/* ------------------------------------------------------------------ */

enum { ORC_OK = 0, ORC_INVALID_TRANSITION = 1, ORC_FLAG_OVERLAP = 2, ORC_BAD_LENGTH = 3 };

/* Column integers of one match_substrs call, each of length M (per def: D x M):
 *   enable, character                      lib.rs:339-348
 *   state, substr_id                       lib.rs:388-395,404-418
 *   start_enable                           lib.rs:482-493
 *   end_enable  (row M-1 is never assigned by the reference, lib.rs:501; reported as 0)
 *   masked_char, masked_substr_id          lib.rs:740-764 (= AssignedRegexResult.masked_characters / .all_substr_ids)
 * Any output pointer may be NULL.
 * info[0..4] = {def,pos,state,char} for ORC_INVALID_TRANSITION; info[1] = row for ORC_FLAG_OVERLAP;
 * info[4] = accept bitmask (bit d: state at row n == accepted_state_val, the value the
 * "is_accepted" assert at lib.rs:427-457 forces) when n < M, else computed from s[n] anyway.
 * The field gates are restated on integers with and=a*b, not=1-a, select(a,b,sel)=sel?a:b
 * (halo2-base v0.2.2 FlexGateConfig).  If a per-row sum of is_start or is_end flags exceeds 1
 * (two defs flag the same row) `not` leaves {0,1} in the field; that is out of contract for
 * the chip and reported as ORC_FLAG_OVERLAP with the lowest such row. */
int orc_match_substrs(const orc *o, const uint8_t *characters, size_t n, size_t M,
                      uint64_t *enable, uint64_t *character,
                      uint64_t *state, uint64_t *substr_id, uint64_t *start_enable, uint64_t *end_enable,
                      uint64_t *masked_char, uint64_t *masked_substr_id, uint64_t *info) {
    const size_t D = o->n_defs;
    if (n > M) return ORC_BAD_LENGTH;
    uint64_t *states = malloc(sizeof(uint64_t) * D * (n + 1));
    uint64_t *sids = malloc(sizeof(uint64_t) * (D * n + 1));
    uint8_t *is_starts = malloc(D * (n + 1));
    uint8_t *is_ends = malloc(D * (n + 1));
    int rc = orc_derive_states(o, characters, n, states, info);             /* lib.rs:316 */
    if (rc) { free(states); free(sids); free(is_starts); free(is_ends); return ORC_INVALID_TRANSITION; }
    orc_derive_substr_ids(o, states, n, sids);                              /* lib.rs:317 */
    orc_derive_is_start_end(o, states, sids, n, is_starts, is_ends);        /* lib.rs:318 */

    uint64_t *en = malloc(sizeof(uint64_t) * M);
    for (size_t i = 0; i < M; i++) {                                        /* lib.rs:339-348 */
        en[i] = i < n ? 1 : 0;
        if (enable) enable[i] = en[i];
        if (character) character[i] = i < n ? characters[i] : 0;
    }
    uint64_t *a_sid = calloc(M, sizeof(uint64_t));                          /* assigned_substr_ids, lib.rs:377-379 */
    uint64_t *a_st = calloc(M + 1, sizeof(uint64_t));                       /* assigned_is_start,   lib.rs:380-382 */
    uint64_t *a_en = calloc(M + 1, sizeof(uint64_t));                       /* assigned_is_end,     lib.rs:383-385 */
    uint64_t accept = 0;

    for (size_t d = 0; d < D; d++) {                                        /* lib.rs:387 */
        const regex_defs *defs = &o->defs[d];
        const uint64_t *s = states + d * (n + 1);
        if (s[n] == defs->accepted_state_val && d < 56) accept |= 1ull << d;
        for (size_t idx = 0; idx < M; idx++) {
            uint64_t state_val, sid_val; uint8_t st, e;
            if (idx < n) {                                                  /* lib.rs:388-403 */
                state_val = s[idx]; sid_val = sids[d * n + idx];
                st = is_starts[d * (n + 1) + idx]; e = is_ends[d * (n + 1) + idx];
            } else if (idx == n) {                                          /* lib.rs:406-411 */
                state_val = s[idx]; sid_val = 0;
                st = is_starts[d * (n + 1) + idx]; e = is_ends[d * (n + 1) + idx];
            } else {                                                        /* lib.rs:412-414 */
                state_val = defs->largest_state_val + 1; sid_val = 0; st = 0; e = 0;
            }
            if (state) state[d * M + idx] = state_val;                      /* lib.rs:419-425 */
            if (substr_id) substr_id[d * M + idx] = sid_val;                /* lib.rs:459-465 */
            a_sid[idx] += sid_val;                                          /* lib.rs:467-471 */
            if (start_enable) start_enable[d * M + idx] = en[idx] * st;     /* lib.rs:482-493 */
            a_st[idx] += st;                                                /* lib.rs:494-498 */
            /* lib.rs:501-519: for idx in 0..M-1 { is_end = is_end_values[idx+1]; end_enable[idx] = enable[idx]*is_end;
             * assigned_is_end[idx+1] += is_end }  — written here from the (idx+1) side. */
            if (idx >= 1) {
                if (end_enable) end_enable[d * M + idx - 1] = en[idx - 1] * e;
                a_en[idx] += e;
            }
        }
        if (end_enable && M >= 1) end_enable[d * M + M - 1] = 0;            /* never assigned by the reference */
    }

    rc = ORC_OK;
    for (size_t idx = 0; idx <= M; idx++) {
        if (a_st[idx] > 1 || a_en[idx] > 1) { rc = ORC_FLAG_OVERLAP; if (info) info[1] = idx; break; }
    }
    if (rc == ORC_OK) {
        uint8_t *start_mask = malloc(M ? M : 1), *end_mask = malloc(M ? M : 1);
        uint64_t last = 0;
        for (size_t idx = 0; idx < M; idx++) {                              /* lib.rs:598-645 */
            uint64_t pre = idx == 0 ? 0 : a_sid[idx - 1];
            uint64_t is_changed = pre != a_sid[idx];
            uint64_t is_set = a_st[idx] * is_changed;
            uint64_t is_reset = (1 - a_st[idx]) * a_en[idx] * is_changed;
            uint64_t nm = is_set ? 1 : last;                                /* select(1, last, is_set) */
            nm = is_reset ? 0 : nm;                                         /* select(0, new, is_reset) */
            start_mask[idx] = (uint8_t)nm; last = nm;
        }
        last = 0;
        for (size_t idx = 0; idx < M; idx++) {                              /* lib.rs:663-714 */
            uint64_t pre = idx == 0 ? 0 : a_sid[M - idx];
            uint64_t is_changed = pre != a_sid[M - 1 - idx];
            uint64_t is_set = a_en[M - idx] * is_changed;
            uint64_t is_reset = (1 - a_en[M - idx]) * a_st[M - idx] * is_changed;
            uint64_t nm = is_set ? 1 : last;
            nm = is_reset ? 0 : nm;
            end_mask[M - 1 - idx] = (uint8_t)nm; last = nm;                 /* end_mask.reverse(), lib.rs:714 */
        }
        for (size_t idx = 0; idx < M; idx++) {                              /* lib.rs:740-764 */
            uint64_t mask = (uint64_t)start_mask[idx] * end_mask[idx];
            if (masked_char) masked_char[idx] = mask * (idx < n ? characters[idx] : 0);
            if (masked_substr_id) masked_substr_id[idx] = mask * a_sid[idx];
        }
        free(start_mask); free(end_mask);
    }
    if (info) info[4] = accept;
    free(states); free(sids); free(is_starts); free(is_ends); free(en); free(a_sid); free(a_st); free(a_en);
    return rc;
}

/* ------------------------------------------------------------------ */
/* compact witness + batch driver                     */
/* ------------------------------------------------------------------ */

/* status word shared with the HIP path (include/hrx.h):
 *   bits 0..7  code
 *   code 0: bits 8..39 accept mask (bit d: state at row n == accepted_state_val of def d; up to 32 defs, HRX_MAX_DEFS)
 *   code 1: bits 8..15 def, 16..23 char, 24..39 state, 40..63 position
 *   code 2: bits 40..63 row
 */
static uint64_t pack_status(int rc, const uint64_t *info) {
    switch (rc) {
    case ORC_OK: return (info[4] & 0xffffffffull) << 8;
    case ORC_INVALID_TRANSITION:
        return 1ull | (info[0] & 0xff) << 8 | (info[3] & 0xff) << 16 | (info[2] & 0xffff) << 24 | (info[1] & 0xffffff) << 40;
    case ORC_FLAG_OVERLAP: return 2ull | (info[1] & 0xffffff) << 40;
    default: return 3ull;
    }
}

/* One string -> compact records.  records: M x D u32 {state:u16 | substr_id:u8<<16 | flags:u8<<24},
 * flags bit0 = start_enable, bit1 = end_enable; masked: M u16 {masked_char | masked_substr_id<<8}. */
uint64_t orc_witness_one(const orc *o, const uint8_t *chars, size_t n, size_t M,
                         uint32_t *records, uint16_t *masked, uint64_t *scratch /* (4*D+2)*M u64 */) {
    const size_t D = o->n_defs;
    uint64_t *state = scratch, *sid = state + D * M, *se = sid + D * M, *ee = se + D * M;
    uint64_t *mc = ee + D * M, *ms = mc + M;
    uint64_t info[5] = {0, 0, 0, 0, 0};
    int rc = orc_match_substrs(o, chars, n, M, NULL, NULL, state, sid, se, ee, mc, ms, info);
    if (rc == ORC_OK) {
        for (size_t r = 0; r < M; r++) {
            for (size_t d = 0; d < D; d++)
                records[r * D + d] = (uint32_t)(state[d * M + r] & 0xffff) | (uint32_t)(sid[d * M + r] & 0xff) << 16 |
                                     (uint32_t)(se[d * M + r] & 1) << 24 | (uint32_t)(ee[d * M + r] & 1) << 25;
            masked[r] = (uint16_t)((mc[r] & 0xff) | (ms[r] & 0xff) << 8);
        }
    }
    return pack_status(rc, info);
}

/* Batch of strings, single thread, string-major buffers (same layout as the HIP path). */
void orc_witness_batch(const orc *o, const uint8_t *chars, size_t stride, const uint32_t *lens, size_t B, size_t M,
                       uint32_t *records, uint16_t *masked, uint64_t *status) {
    const size_t D = o->n_defs;
    uint64_t *scratch = malloc(sizeof(uint64_t) * (4 * D + 2) * (M ? M : 1));
    for (size_t b = 0; b < B; b++)
        status[b] = orc_witness_one(o, chars + b * stride, lens[b], M, records + b * M * D, masked + b * M, scratch);
    free(scratch);
}

/* The same batch spread over `threads` host threads, one string per task (SURVEY §8d "all host cores" baseline). */
typedef struct { const orc *o; const uint8_t *chars; size_t stride; const uint32_t *lens; size_t B, M, lo, hi;
                 uint32_t *records; uint16_t *masked; uint64_t *status; } mt_job;
static void *mt_worker(void *p) {
    mt_job *j = p;
    const size_t D = j->o->n_defs;
    uint64_t *scratch = malloc(sizeof(uint64_t) * (4 * D + 2) * (j->M ? j->M : 1));
    for (size_t b = j->lo; b < j->hi; b++)
        j->status[b] = orc_witness_one(j->o, j->chars + b * j->stride, j->lens[b], j->M, j->records + b * j->M * D,
                                       j->masked + b * j->M, scratch);
    free(scratch);
    return NULL;
}
void orc_witness_batch_mt(const orc *o, const uint8_t *chars, size_t stride, const uint32_t *lens, size_t B, size_t M,
                          uint32_t *records, uint16_t *masked, uint64_t *status, size_t threads) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t tid[256]; mt_job job[256];
    for (size_t t = 0; t < threads; t++) {
        job[t] = (mt_job){o, chars, stride, lens, B, M, B * t / threads, B * (t + 1) / threads, records, masked, status};
        pthread_create(&tid[t], NULL, mt_worker, &job[t]);
    }
    for (size_t t = 0; t < threads; t++) pthread_join(tid[t], NULL);
}

/* ------------------------------------------------------------------ */
/* "best CPU" variant: dense fused table, no hashing       */
/* ------------------------------------------------------------------ */
/* Not a restatement of the reference's data structures: one u32 per (state, byte) holding everything lib.rs:804-888
 * derives from that pair — next | substr_id<<16 | is_start<<24 | is_end<<25 | valid<<31 — filled by asking the
 * reference-shaped containers above, then one table read per (row, def).  tests/test_oracle_golden.py checks it
 * equal to orc_witness_batch; bench.py times it as the strongest CPU line next to the port. */
typedef struct { uint32_t *tab[32]; uint64_t first[32], accepted[32], dummy[32]; size_t n_defs; } orc_dense;

void orc_dense_free(orc_dense *t) { if (!t) return; for (size_t d = 0; d < t->n_defs; d++) free(t->tab[d]); free(t); }

orc_dense *orc_dense_new(const orc *o) {
    if (o->n_defs > 32) return NULL;
    orc_dense *t = calloc(1, sizeof(orc_dense));
    t->n_defs = o->n_defs;
    uint64_t off = 1;
    for (size_t d = 0; d < o->n_defs; d++) {
        const regex_defs *defs = &o->defs[d];
        const size_t S = (size_t)defs->largest_state_val + 1;
        t->first[d] = defs->first_state_val; t->accepted[d] = defs->accepted_state_val; t->dummy[d] = S;
        t->tab[d] = calloc(S * 256, sizeof(uint32_t));
        for (size_t s = 0; s < S; s++)
            for (unsigned c = 0; c < 256; c++) {
                const lookup_slot *e = map_find(&defs->state_lookup, (uint8_t)c, s);
                if (!e) continue;
                uint32_t v = 0x80000000u | (uint32_t)(e->next & 0xffff);
                for (size_t j = 0; j < defs->n_substrs; j++)
                    if (set_contains(&defs->substrs[j].valid_state_transitions, s, e->next)) {
                        v |= (uint32_t)((off + j) & 0xff) << 16;
                        if (vec_contains(&defs->substrs[j].start_states, s)) v |= 1u << 24;
                        if (vec_contains(&defs->substrs[j].end_states, e->next)) v |= 1u << 25;
                        break;
                    }
                t->tab[d][s * 256 + c] = v;
            }
        off += defs->n_substrs;
    }
    return t;
}

static uint64_t dense_one(const orc_dense *t, const uint8_t *chars, size_t n, size_t M, uint32_t *records, uint16_t *masked,
                          uint8_t *sum /* 3*(M+1): SID, ST, EN */) {
    const size_t D = t->n_defs;
    if (n > M) return 3ull;
    uint8_t *SID = sum, *ST = SID + M + 1, *EN = ST + M + 1;
    memset(sum, 0, 3 * (M + 1));
    uint64_t accept = 0;
    for (size_t d = 0; d < D; d++) {
        const uint32_t *tab = t->tab[d];
        uint32_t s = (uint32_t)t->first[d];
        size_t r = 0;
        for (; r < n; r++) {
            uint32_t e = tab[(size_t)s * 256 + chars[r]];
            if (!(e & 0x80000000u))
                return 1ull | (uint64_t)d << 8 | (uint64_t)chars[r] << 16 | (uint64_t)(s & 0xffff) << 24 | (uint64_t)(r & 0xffffff) << 40;
            records[r * D + d] = s | (e & 0x01ff0000u);                     /* state, sid, start_enable */
            SID[r] += (uint8_t)(e >> 16); ST[r] += (uint8_t)(e >> 24 & 1);
            if (r + 1 < M) {                                                /* lib.rs:501-519 stops at M-2: row M-1 never set */
                records[r * D + d] |= e & 0x02000000u;                      /* end_enable[r] = is_end[r+1] */
                EN[r + 1] += (uint8_t)(e >> 25 & 1);
            }
            s = e & 0xffff;
        }
        if (s == t->accepted[d]) accept |= 1ull << d;
        if (r < M) records[r++ * D + d] = s;
        for (; r < M; r++) records[r * D + d] = (uint32_t)t->dummy[d];
    }
    for (size_t r = 0; r <= M; r++)
        if (ST[r] > 1 || EN[r] > 1) return 2ull | (uint64_t)(r & 0xffffff) << 40;
    uint8_t m = 0;
    for (size_t r = 0; r < M; r++) {                                        /* forward start_mask, lib.rs:598-645 */
        int chg = (r ? SID[r - 1] : 0) != SID[r];
        if (chg && ST[r]) m = 1; else if (chg && EN[r]) m = 0;
        masked[r] = m;
    }
    m = 0;
    for (size_t r = M; r-- > 0;) {                                          /* backward end_mask, lib.rs:663-714 */
        int chg = (r + 1 < M ? SID[r + 1] : 0) != SID[r];
        if (chg && EN[r + 1]) m = 1; else if (chg && ST[r + 1]) m = 0;
        masked[r] = (masked[r] & m) ? (uint16_t)((r < n ? chars[r] : 0) | SID[r] << 8) : 0;
    }
    return accept << 8;
}

typedef struct { const orc_dense *t; const uint8_t *chars; size_t stride; const uint32_t *lens; size_t M, lo, hi;
                 uint32_t *records; uint16_t *masked; uint64_t *status; } dense_job;
static void *dense_worker(void *p) {
    dense_job *j = p;
    const size_t D = j->t->n_defs;
    uint8_t *sum = malloc(3 * (j->M + 1));
    for (size_t b = j->lo; b < j->hi; b++)
        j->status[b] = dense_one(j->t, j->chars + b * j->stride, j->lens[b], j->M, j->records + b * j->M * D, j->masked + b * j->M, sum);
    free(sum);
    return NULL;
}
void orc_dense_witness_batch(const orc_dense *t, const uint8_t *chars, size_t stride, const uint32_t *lens, size_t B, size_t M,
                             uint32_t *records, uint16_t *masked, uint64_t *status, size_t threads) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t tid[256]; dense_job job[256];
    for (size_t i = 0; i < threads; i++) {
        job[i] = (dense_job){t, chars, stride, lens, M, B * i / threads, B * (i + 1) / threads, records, masked, status};
        if (threads == 1) { dense_worker(&job[0]); return; }
        pthread_create(&tid[i], NULL, dense_worker, &job[i]);
    }
    for (size_t i = 0; i < threads; i++) pthread_join(tid[i], NULL);
}

/* ------------------------------------------------------------------ */
/* RegexTableConfig::load rows — src/table.rs:61-198                   */
/* ------------------------------------------------------------------ */

static int cmp_line_idx(const void *a, const void *b) {
    const lookup_slot *x = *(const lookup_slot *const *)a, *y = *(const lookup_slot *const *)b;
    return x->line_idx < y->line_idx ? -1 : x->line_idx > y->line_idx;
}

/* Transition table of def d: rows of (char, cur_state, next_state, substr_id).
 * Row 0 = (0, dummy, dummy, 0) (table.rs:101), then one row per state_lookup entry sorted by
 * line index (table.rs:103-108) with the first matching substr's id (table.rs:110-120).
 * substr_id_offset as in RegexVerifyConfig::load (lib.rs:780-783).  Returns the row count. */
size_t orc_table_transition_rows(const orc *o, size_t d, uint64_t *rows, size_t cap_rows) {
    const regex_defs *defs = &o->defs[d];
    uint64_t off = 1;
    for (size_t i = 0; i < d; i++) off += o->defs[i].n_substrs;
    uint64_t dummy = defs->largest_state_val + 1;                           /* table.rs:67 */
    size_t nrows = 1 + defs->state_lookup.len;
    if (!rows || cap_rows < nrows) return nrows;
    rows[0] = 0; rows[1] = dummy; rows[2] = dummy; rows[3] = 0;
    const lookup_slot **ptrs = malloc(sizeof(void *) * (defs->state_lookup.len + 1));
    size_t k = 0;
    for (size_t i = 0; i < defs->state_lookup.cap; i++) if (defs->state_lookup.slots[i].used) ptrs[k++] = &defs->state_lookup.slots[i];
    qsort(ptrs, k, sizeof(void *), cmp_line_idx);
    for (size_t i = 0; i < k; i++) {
        uint64_t sid = 0;
        for (size_t j = 0; j < defs->n_substrs; j++)
            if (set_contains(&defs->substrs[j].valid_state_transitions, ptrs[i]->state, ptrs[i]->next)) { sid = off + j; break; }
        uint64_t *r = rows + 4 * (i + 1);
        r[0] = ptrs[i]->ch; r[1] = ptrs[i]->state; r[2] = ptrs[i]->next; r[3] = sid;
    }
    free(ptrs);
    return nrows;
}

/* Endpoint table of def d: rows of (substr_id, start_state, end_state) — table.rs:126-196. */
size_t orc_table_endpoint_rows(const orc *o, size_t d, uint64_t *rows, size_t cap_rows) {
    const regex_defs *defs = &o->defs[d];
    uint64_t off = 1;
    for (size_t i = 0; i < d; i++) off += o->defs[i].n_substrs;
    uint64_t dummy = defs->largest_state_val + 1;
    size_t nrows = 1;
    for (size_t j = 0; j < defs->n_substrs; j++) nrows += defs->substrs[j].start_states.len + defs->substrs[j].end_states.len;
    if (!rows || cap_rows < nrows) return nrows;
    size_t k = 0;
    rows[0] = 0; rows[1] = dummy; rows[2] = dummy; k = 1;                    /* table.rs:130-148 */
    for (size_t j = 0; j < defs->n_substrs; j++) {
        for (size_t i = 0; i < defs->substrs[j].start_states.len; i++, k++) {  /* table.rs:151-171 */
            rows[3 * k] = off + j; rows[3 * k + 1] = defs->substrs[j].start_states.v[i]; rows[3 * k + 2] = dummy;
        }
        for (size_t i = 0; i < defs->substrs[j].end_states.len; i++, k++) {    /* table.rs:172-192 */
            rows[3 * k] = off + j; rows[3 * k + 1] = dummy; rows[3 * k + 2] = defs->substrs[j].end_states.v[i];
        }
    }
    return nrows;
}

/* ---------------------------------------------------------------------------------------------------------------
 * F::from(u64) for F = halo2curves bn256::Fr — the value inside `Value::known(F::from(v))` at
 * src/lib.rs:342-347, 390-417.  The arithmetic lives in a third-party crate that is NOT in /root/reference:
 * halo2curves, pulled in through halo2-base v0.2.2 (axiom-crypto/halo2-lib @ 9860acc, Cargo.toml:12-15).  Restated from
 * its published source (src/bn256/fr.rs + the field_arithmetic! macro): an element is 4 little-endian u64 limbs in
 * Montgomery form, `From<u64>` is `Fr([v, 0, 0, 0]) * R2`, and `mul` is a schoolbook 4x4 product followed by
 * `montgomery_reduce` with INV = -r^-1 mod 2^64.  Constants as published there:
 *   MODULUS = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
 *   R  = [ac96341c4ffffffb, 36fc76959f60cd29, 666ea36f7879462e, 0e0a77c19a07df2f]   (2^256 mod r; Fr::one())
 *   R2 = [1bb8e645ae216da7, 53fe3ab1e35c59e3, 8c49833d53bb8085, 0216d0b17f4e44a5]   (2^512 mod r)
 *   INV = 0xc2e1f593efffffff
 * Parity status of this function: PARTIAL — the reference's own tests reach it only through MockProver
 * (lib.rs:1052-1059 compares F::from(expected) with assigned cells); tests/test_fr.py pins it to the constants above
 * (from(1) == R, R2 == R*R mod r, INV*r == -1 mod 2^64) and to exact big-integer arithmetic (v * 2^256 mod r).
 * The product computes the same value by a different route (csrc/hrx_fr.h).
 * --------------------------------------------------------------------------------------------------------------- */
static const uint64_t FR_MODULUS[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t FR_R2[4] = {0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull, 0x0216d0b17f4e44a5ull};
static const uint64_t FR_INV = 0xc2e1f593efffffffull;

typedef unsigned __int128 orc_u128;
static uint64_t fr_mac(uint64_t a, uint64_t b, uint64_t c, uint64_t *carry) { /* a + b*c + carry */
    const orc_u128 t = (orc_u128)a + (orc_u128)b * c + *carry;
    *carry = (uint64_t)(t >> 64);
    return (uint64_t)t;
}
static uint64_t fr_adc(uint64_t a, uint64_t b, uint64_t *carry) {
    const orc_u128 t = (orc_u128)a + b + *carry;
    *carry = (uint64_t)(t >> 64);
    return (uint64_t)t;
}

/* Fr::mul: 8-limb product, then montgomery_reduce, then one conditional subtraction of the modulus */
static void fr_mul(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) {
    uint64_t t[8] = {0};
    for (int i = 0; i < 4; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < 4; j++) t[i + j] = fr_mac(t[i + j], a[i], b[j], &carry);
        t[i + 4] = carry;
    }
    uint64_t carry2 = 0;
    for (int i = 0; i < 4; i++) {
        const uint64_t k = t[i] * FR_INV;
        uint64_t carry = 0;
        (void)fr_mac(t[i], k, FR_MODULUS[0], &carry);
        for (int j = 1; j < 4; j++) t[i + j] = fr_mac(t[i + j], k, FR_MODULUS[j], &carry);
        t[i + 4] = fr_adc(t[i + 4], carry2, &carry);
        carry2 = carry;
    }
    /* result = t[4..8] (+ carry2 * 2^256) - (modulus if >= modulus) */
    uint64_t r[4], borrow = 0;
    for (int i = 0; i < 4; i++) {
        const orc_u128 d = (orc_u128)t[4 + i] - FR_MODULUS[i] - borrow;
        r[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1u;
    }
    const int ge = carry2 || !borrow;
    for (int i = 0; i < 4; i++) out[i] = ge ? r[i] : t[4 + i];
}

void orc_fr_from_u64(uint64_t v, uint64_t out[4]) {
    const uint64_t a[4] = {v, 0, 0, 0};
    fr_mul(a, FR_R2, out);
}
void orc_fr_constants(uint64_t modulus[4], uint64_t r2[4], uint64_t *inv) {
    memcpy(modulus, FR_MODULUS, 32);
    memcpy(r2, FR_R2, 32);
    *inv = FR_INV;
}
