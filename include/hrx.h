/*
 * hrx.h — C ABI of the MI355X-native batched DFA witness generator for halo2-regex.
 *
 * The reference (zkemail/halo2-regex) has no FFI: the seam this library replaces is
 * Rust-internal — the three pub(crate) methods RegexVerifyConfig::derive_states /
 * derive_substr_ids / derive_is_start_end (src/lib.rs:804-888) called at the top of
 * RegexVerifyConfig::match_substrs (src/lib.rs:316-318), the integer content of the
 * reveal-mask scan in match_substrs (src/lib.rs:593-764), the data model and text
 * parsers of src/defs.rs, and the fixed-table rows of RegexTableConfig::load
 * (src/table.rs:61-198).  Every entry point below cites the reference interface it
 * stands in for.  INTEGRATION.md shows the Rust binding a maintainer would add.
 *
 * Conventions: plain pointers and sizes, caller-owned buffers, integer status
 * returns (0 = HRX_OK), nothing unwinds.  Text of the last error: hrx_last_error().
 * Batches run on a gfx950 device: the device-pointer entry points fail with HRX_ERR_HIP without one.  The library's one
 * host-side compute path is its native small-batch walk (the same lane algorithm on a host core): the reference-shaped
 * single-string entry points and host-buffer batches below the context's threshold take it — match_substrs hands over ONE
 * string per call (src/lib.rs:316-318) and a GPU launch cannot beat a host core on ~1000 rows.
 */
#ifndef HRX_H
#define HRX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hrx_defs hrx_defs; /* Vec<RegexDefs> (src/defs.rs:17-22) + the dense fused tables built from it */
typedef struct hrx_ctx hrx_ctx;   /* one device: tables resident in HBM, a stream, staging buffers */

enum {
    HRX_OK = 0,
    HRX_ERR_PARSE = 1,              /* where AllstrRegexDef/SubstrRegexDef::read_from_reader panics (defs.rs:85-91, 219-225) */
    HRX_ERR_BOUNDS = 2,             /* state / substr-id ranges the compact record or the LDS table cannot hold */
    HRX_ERR_ARG = 3,
    HRX_ERR_HIP = 4,
    HRX_ERR_STATE = 5,              /* call order (e.g. push after finalize) */
    HRX_ERR_INVALID_TRANSITION = 6, /* single-string API: the reference's panic at src/lib.rs:817 */
    HRX_ERR_OUT_OF_CONTRACT = 7,    /* single-string API: two defs flag the same row or n > max_chars_size */
    HRX_ERR_IO = 8
};

/* ------------------------------------------------------------------ */
/* Data model — src/defs.rs                                            */
/* ------------------------------------------------------------------ */

int hrx_defs_create(hrx_defs **out);
void hrx_defs_destroy(hrx_defs *defs);

/* AllstrRegexDef::read_from_reader (defs.rs:75-110) on an in-memory text; begins a new
 * RegexDefs { allstr, substrs: vec![] }.  Line 0 first state, line 1 accepted state,
 * line 2 largest state, then "cur next char" per line; the char is taken `as u8`. */
int hrx_defs_push_allstr_text(hrx_defs *defs, const char *text, size_t len);
/* AllstrRegexDef::read_from_text (defs.rs:54-58) */
int hrx_defs_push_allstr_file(hrx_defs *defs, const char *path);
/* SubstrRegexDef::read_from_reader (defs.rs:209-265); appended to the last RegexDefs.substrs. */
int hrx_defs_push_substr_text(hrx_defs *defs, const char *text, size_t len);
/* SubstrRegexDef::read_from_text (defs.rs:184-188) */
int hrx_defs_push_substr_file(hrx_defs *defs, const char *path);
/* Already-parsed forms for a Rust caller that holds the structs:
 * AllstrRegexDef { state_lookup, first_state_val, accepted_state_val, largest_state_val } (defs.rs:26-36);
 * entry i is state_lookup[(chr[i], cur[i])] = (line_idx[i], next[i]). */
int hrx_defs_push_allstr(hrx_defs *defs, uint64_t first_state_val, uint64_t accepted_state_val,
                         uint64_t largest_state_val, size_t n_transitions, const uint64_t *cur,
                         const uint64_t *next, const uint8_t *chr, const uint64_t *line_idx);
/* SubstrRegexDef::new (defs.rs:147-163); max_length / min_position / max_position are unused by the chip. */
int hrx_defs_push_substr(hrx_defs *defs, size_t n_pairs, const uint64_t *pair_cur, const uint64_t *pair_next,
                         size_t n_start, const uint64_t *start_states, size_t n_end, const uint64_t *end_states);
/* Validate and build the dense fused (state,char) tables.  Required before any call below.
 * regex_defs is a Vec of any length in the reference (src/lib.rs:112; loops at :387, :806, :828, :855): up to HRX_MAX_DEFS
 * RegexDefs per config.  Up to three defs are walked side by side by one kernel launch; four to eight defs of at most 32 byte classes each (position-major
 * outputs) by ONE def-parallel launch — a walker wave per def over class-indexed tables, a combiner wave — at 0.66-0.70 of the HBM peak (nine defs and more: one such launch per group of up to eight defs + the combine launch, 0.58-0.65); a def of more than 32 byte classes puts its config into passes
 * over consecutive groups of defs (each group's tables LDS-resident); the per-row sums over all defs (reveal masks, flag
 * overlap) and the merged status are formed by the last pass itself from 80-byte tile summaries the earlier passes leave
 * (position-major outputs, up to four groups) or by a combine launch — same buffers, same results; with position-major outputs
 * the passes write the caller's record planes directly and the extra cost is ~2.5 bytes per row and earlier group.  String-major
 * outputs of more than three defs: four and five defs of at most 32 byte classes each, row counts in multiples of 16, come straight out of the
 * def-parallel launch (0.50-0.68 of peak: the records meet in LDS sub-tiles, a storer wave writes the caller's [B][pitch][D]); otherwise row counts
 * in multiples of 8 are produced position-major in context scratch and transposed (~0.2 of peak), and other row counts are copied into place by
 * the combine launch and are several times slower: ask for position-major buffers there.
 * HRX_MP_COMBINE=1 in the environment of hrx_ctx_create keeps the separate combine launch. */
#define HRX_MAX_DEFS 32
int hrx_defs_finalize(hrx_defs *defs);

size_t hrx_defs_num_defs(const hrx_defs *defs);
size_t hrx_defs_num_substrs(const hrx_defs *defs, size_t def);
uint64_t hrx_defs_first_state(const hrx_defs *defs, size_t def);
uint64_t hrx_defs_accepted_state(const hrx_defs *defs, size_t def);
uint64_t hrx_defs_largest_state(const hrx_defs *defs, size_t def);
size_t hrx_defs_num_transitions(const hrx_defs *defs, size_t def);
/* substr_id of the first substring of `def`: the offset rule of lib.rs:780-783 / 827,842 */
uint64_t hrx_defs_substr_id_offset(const hrx_defs *defs, size_t def);
/* bytes of LDS the fused tables of all defs occupy */
size_t hrx_defs_table_bytes(const hrx_defs *defs);

/* RegexTableConfig::load (table.rs:61-198), rows as integers in assignment order.
 * transition rows: (char, cur_state, next_state, substr_id), row 0 = (0,dummy,dummy,0), then one row per
 * state_lookup entry sorted by line index.  endpoint rows: (substr_id, start_state, end_state).
 * Returns the number of rows; writes min(rows, cap) of them. */
size_t hrx_table_transition_rows(const hrx_defs *defs, size_t def, uint64_t *rows4, size_t cap_rows);
size_t hrx_table_endpoint_rows(const hrx_defs *defs, size_t def, uint64_t *rows3, size_t cap_rows);

/* ------------------------------------------------------------------ */
/* Device context                                                      */
/* ------------------------------------------------------------------ */

int hrx_device_count(int *count);
/* Uploads the tables to `device` and creates a stream.  The handle may be shared by clones of a
 * RegexVerifyConfig (lib.rs:96 derives Clone); device work of one ctx is serialised internally.  A context owns device
 * scratch that some launches use (the group counter of multi-round batches, the group buffers of configs of more than three
 * defs): it serves ONE stream or captured graph at a time — a launch on another stream first waits on the host for the stream
 * that used the scratch last (HRX_ERR_STATE if that wait is impossible, e.g. inside a stream capture), and a captured graph
 * must not be replayed concurrently with other launches of the same context; use one context per concurrent stream.  Every entry point
 * leaves the caller's current HIP device as it found it.
 * device = HRX_DEVICE_NONE: a host-only context (no HIP call is made): the single-string entry points and
 * hrx_witness_batch_host run the native host walk, device-pointer entry points return HRX_ERR_HIP.
 * HRX_DEBUG_FLAGS (environment; kernel-selection bits for the tests, csrc/hrx_kernel.hpp) is read here, once. */
#define HRX_DEVICE_NONE (-1)
int hrx_ctx_create(const hrx_defs *defs, int device, hrx_ctx **out);
/* A second context of the same config: its own stream, scratch and lock, the same tables (uploaded again: ~100 KiB) and the source's per-context switches
 * (host threshold, placement).  RegexVerifyConfig derives Clone (src/lib.rs:96) and halo2 clones the config per synthesize pass / worker thread: a context
 * serves one stream at a time, so a prover that wants its threads to overlap gives every clone a context of its own (they share the device's measured arena pair,
 * hrx_alloc_output_pair).  device: a GPU index, HRX_DEVICE_NONE, or HRX_DEVICE_SAME (the source's).  The hrx_defs handle the source was made from need not exist any more. */
#define HRX_DEVICE_SAME (-2)
int hrx_ctx_clone(const hrx_ctx *ctx, int device, hrx_ctx **out);
void hrx_ctx_destroy(hrx_ctx *ctx);
int hrx_ctx_device(const hrx_ctx *ctx);
/* Host-buffer batches (hrx_witness_batch_host, hrx_multi_witness_batch_host) of fewer than `rows` witness rows (B x M)
 * are walked on the calling host thread instead of being staged to the device; default HRX_DEFAULT_HOST_THRESHOLD
 * (the measured crossover, NOTES_MEASUREMENTS.md §7c); 0 = always the device.  The single-string entry points below always take the
 * host walk (one GPU lane needs ~50 ns per row, a host core ~3).  Results are identical either way. */
#define HRX_DEFAULT_HOST_THRESHOLD 32768
int hrx_ctx_set_host_threshold(hrx_ctx *ctx, size_t rows);
size_t hrx_ctx_host_threshold(const hrx_ctx *ctx);
/* Per-context choices between variants that compute the same rows (what a linking prover sets instead of environment variables).
 *   HRX_OPT_PMD_COMBINER_WAVE  the def-parallel kernel of two- and three-def configs (batches of at most two groups of 64 strings per CU): 1 = a combiner wave of its own
 *                              per group (sums over the defs, reveal mask, masked rows), 2 = the last def's walker combines, 0 = the library's default
 *   HRX_OPT_HOST_ROUTE         hrx_witness_batch_host on a device context, batches of at least the host threshold:
 *                              HRX_HOST_ROUTE_AUTO (0, default) from 2^22 rows on the FASTEST OF THREE WAYS by this context's own measurements: everything through the device
 *                                (staged, walked, copied back: the copy back over the link bounds it, ~9e9 rows/s at one def), everything on the host cores (the native walk: the host's
 *                                cores and memory bound it), or both at once — the batch split by string index between them in the ratio of the two parts' rates.  The context's calls 0
 *                                and 1 go through the device (0 pays for allocations and is not recorded), 2 on the host cores, 3 and 4 split; from then on the way with the smallest time
 *                                per row, whose figure every call refreshes; every 32nd call re-measures one of the other two if its figure was within 1.5x of the best.  Smaller
 *                                batches: the device; below the host threshold: the host;
 *                              HRX_HOST_ROUTE_DEVICE (1) everything through the device;  HRX_HOST_ROUTE_HOST (2) everything on the host cores.  Results are identical either way.
 *   HRX_OPT_HOST_THREADS       host threads of the native walk (0, default: as many as the calling thread's affinity mask has cores; hrx_multi_create divides them among its shards)
 *   HRX_OPT_HOST_PIPELINE      the device part's transfers: 0 (default) the context times both ways over its first calls and keeps the faster (the pipeline's copies out run at half rate
 *                              on some hosts), 1 pipelined chunk by chunk over two streams, 2 in, walk, out on one stream; HRX_HOST_PIPELINE=1 / 0 in the environment of hrx_ctx_create
 *                              sets 1 / 2 as the context's default
 *   HRX_OPT_PLACE_DRY_LAUNCH   hrx_alloc_output_planes / _for_batch: 1 (default) the best candidate sets by the pairings' score are each launched into and the fastest is kept, 0 the best
 *                              by score is kept unlaunched (a profiler's kernel list then holds the caller's launches only)
 * Applies to later calls; hrx_ctx_clone copies the options.  hrx_ctx_get_option: the value, -1 for an unknown option. */
enum { HRX_OPT_PMD_COMBINER_WAVE = 1, HRX_OPT_HOST_ROUTE = 2, HRX_OPT_HOST_THREADS = 3, HRX_OPT_HOST_PIPELINE = 4, HRX_OPT_PLACE_DRY_LAUNCH = 5 };
enum { HRX_HOST_ROUTE_AUTO = 0, HRX_HOST_ROUTE_DEVICE = 1, HRX_HOST_ROUTE_HOST = 2 };
/* What the context's last hrx_witness_batch_host call did. */
typedef struct hrx_host_route_report {
    int route;                    /* what the call did — 0: split between the device and the host cores, 1: device only, 2: host cores only */
    size_t device_strings, host_strings;
    double device_ms, host_ms;    /* wall time of each part (they run at the same time) */
    double call_ms;
    double device_alone_ns_per_row, host_alone_ns_per_row, split_ns_per_row;   /* the context's figures after the call (0: not measured yet): what HRX_HOST_ROUTE_AUTO picks its way by */
    double device_ns_per_row, host_ns_per_row;   /* ... of the two parts of a split call (both running at once): what the next split is made from */
    int host_threads;
    int device_pipelined;         /* the device part: 1 chunks pipelined over two streams, 0 one stream */
} hrx_host_route_report;
int hrx_ctx_host_route_report(const hrx_ctx *ctx, hrx_host_route_report *out, size_t out_bytes);   /* out_bytes = sizeof(hrx_host_route_report) of the caller's header: the struct may grow at its end */
int hrx_ctx_set_option(hrx_ctx *ctx, int option, long value);
long hrx_ctx_get_option(const hrx_ctx *ctx, int option);
/* thread-local text of the last failing call (any entry point) */
const char *hrx_last_error(void);

/* ------------------------------------------------------------------ */
/* The hot path: witness rows for a batch of strings                    */
/* ------------------------------------------------------------------ */
/*
 * One call = match_substrs' integer content (lib.rs:316-318, 339-348, 387-519, 593-764) for B strings.
 *   chars    B strings, string b at chars + b*stride, stride % 16 == 0, base 16-byte aligned
 *   lens     n_b (bytes of string b), n_b <= M
 *   M        max_chars_size (lib.rs:128): witness rows per string
 *   records  [B][M][D] u32: state_d[r] | substr_id_d[r] << 16 | start_enable_d[r] << 24 | end_enable_d[r] << 25
 *            (the states / substr_ids / start_enable / end_enable advice columns, lib.rs:419-519)
 *   masked   [B][M] u16:   masked_char[r] | masked_substr_id[r] << 8
 *            (AssignedRegexResult.masked_characters / .all_substr_ids, lib.rs:766-771)
 *   status   [B] u64: bits 0..7 code
 *            0 ok: bits 8..39 = accept mask (bit d: state at row n == accepted_state_val of def d, lib.rs:442-457)
 *            1 invalid transition (lib.rs:817): bits 8..15 def, 16..23 char, 24..39 state, 40..63 position;
 *              lowest def, then lowest position, like the reference's loop order
 *            2 two defs flag the same row (out of contract): bits 40..63 row
 *            3 n_b > M
 *            records/masked of a string whose code != 0 are unspecified.
 * All five pointers are DEVICE pointers on ctx's device; `stream` is a hipStream_t (NULL = the HIP null stream).
 * The call is asynchronous on that stream.
 */
int hrx_witness_batch_device(hrx_ctx *ctx, const uint8_t *chars, size_t stride, const uint32_t *lens, size_t B,
                             size_t M, uint32_t *records, uint16_t *masked, uint64_t *status, void *stream);
/* Same with pitched outputs: string b's rows start at records + b*rec_pitch*D and masked + b*msk_pitch (pitches in rows,
 * >= M, multiples of 8 when M is).  A pitch that is not a power of two keeps the chip-wide write front off a subset of the
 * HBM channels (NOTES_MEASUREMENTS.md §4); hrx_recommended_pitches gives values for a given M (and an input stride). */
int hrx_witness_batch_device_pitched(hrx_ctx *ctx, const uint8_t *chars, size_t stride, const uint32_t *lens, size_t B,
                                     size_t M, uint32_t *records, size_t rec_pitch, uint16_t *masked, size_t msk_pitch,
                                     uint64_t *status, void *stream);
void hrx_recommended_pitches(size_t M, size_t *rec_pitch, size_t *msk_pitch, size_t *chars_stride);
/* Output layouts of the device entry point below.
 *   HRX_LAYOUT_STRING_MAJOR   records [B][M][D], masked [B][M]                       (hrx_witness_batch_device)
 *   HRX_LAYOUT_POSITION_MAJOR records [ceil(M/4)][D][B][4], masked [ceil(M/8)][B][8] for B <= HRX_PM_BLOCK (65536):
 *       record of (string b, row r, def d) at ((r/4*D + d)*B + b)*4 + r%4;  masked of (b, r) at (r/8*B + b)*8 + r%8.
 *       Four rows of one string and def are 16 contiguous bytes and consecutive strings are adjacent, so with one GPU
 *       lane per string every store instruction writes one contiguous 1-KiB run of full lines, and the whole device
 *       writes into one compact slab at a time — the layout the HBM write path rewards (NOTES_MEASUREMENTS.md §4); rows >= M of
 *       the last quad/octet are unspecified.
 *       Larger batches are BLOCKED: strings [k*HRX_PM_BLOCK, (k+1)*HRX_PM_BLOCK) form block k, each block is a complete
 *       array of the above shape over its own nb = min(HRX_PM_BLOCK, B - k*HRX_PM_BLOCK) strings, blocks back to back:
 *       with k = b / HRX_PM_BLOCK, b' = b % HRX_PM_BLOCK,
 *         record at k*HRX_PM_BLOCK*ceil(M/4)*D*4 + ((r/4*D + d)*nb + b')*4 + r%4,
 *         masked at k*HRX_PM_BLOCK*ceil(M/8)*8   + (r/8*nb + b')*8 + r%8
 *       (the distance between a string's consecutive quads stays <= 1 MiB per def whatever B is: 14-28 % faster than one
 *       array over 262144 strings).  Buffer sizes: hrx_position_major_sizes.  The values equal the string-major ones.
 *   HRX_LAYOUT_INPUT_POSITION_MAJOR (or-ed with HRX_LAYOUT_POSITION_MAJOR): the input is chunked and blocked the same
 *       way, per block chars [stride/16][nb][16]: byte i of string b at k*HRX_PM_BLOCK*stride + ((i/16)*nb + b')*16 + i%16
 *       — what a caller that assembles the batch itself should write; `stride` is then only the per-string capacity
 *       (stride % 16 == 0, >= every n_b). */
#define HRX_PM_BLOCK 65536
enum { HRX_LAYOUT_STRING_MAJOR = 0, HRX_LAYOUT_POSITION_MAJOR = 1, HRX_LAYOUT_INPUT_POSITION_MAJOR = 2,
       HRX_LAYOUT_RECORD_PLANES = 4 /* hrx_describe_launch / hrx_ctx_describe_launch only, or-ed with HRX_LAYOUT_POSITION_MAJOR: the launch of hrx_witness_batch_device_planes */ };
int hrx_witness_batch_device_layout(hrx_ctx *ctx, int layout, const uint8_t *chars, size_t stride, const uint32_t *lens,
                                    size_t B, size_t M, uint32_t *records, uint16_t *masked, uint64_t *status, void *stream);
void hrx_position_major_sizes(size_t B, size_t M, size_t D, size_t *records_u32, size_t *masked_u16);
/* RECORD PLANES — HRX_LAYOUT_POSITION_MAJOR with every def's records in a buffer of its own: plane d holds the states / substr_ids / start_enable / end_enable
 * columns of regex_defs[d] (src/lib.rs:387-519 fills them per def), laid out like the position-major records of a one-def config — per block of HRX_PM_BLOCK
 * strings [ceil(M/4)][nb][4] u32, blocks back to back: record of (string b, row r) of def d at record_planes[d][k*HRX_PM_BLOCK*ceil(M/4)*4 + ((r/4)*nb + b')*4 + r%4];
 * masked, status, chars, lens and `layout` (HRX_LAYOUT_POSITION_MAJOR, optionally | HRX_LAYOUT_INPUT_POSITION_MAJOR) as in hrx_witness_batch_device_layout.
 * Why: a launch of D defs writes 4 D of its 4 D + 2 output bytes per row into the records; as ONE allocation they lie in one class of the MI355X's physical
 * address space and the launch runs at what one class absorbs, as separately placed planes (hrx_alloc_output_planes) its D + 1 write streams spread over the
 * classes: three defs x 32768 x 32768 rows 0.78-0.80 of the HBM peak against 0.70-0.72, two defs x 2^20 x 2048 rows 0.81 against 0.75-0.78 (DESIGN.md §6,
 * profiles/r06_probes/).
 * n_planes = the config's number of defs.  ONE def may also come in TWO ROW STRIPES (n_planes = 2): quad q = r/4 of a string lies in record_planes[q % 2] at slot q / 2,
 * each stripe per block [ceil(ceil(M/4)/2)][nb][4] — record of (b, r) at record_planes[(r/4) % 2][k*HRX_PM_BLOCK*ceil(ceil(M/4)/2)*4 + ((r/8)*nb + b')*4 + r%4] — so that
 * a one-def launch's record bytes spread over two classes beside the masked rows' third (sizes: hrx_position_major_stripe_sizes).
 * Configs that run as one launch (up to three defs, or four to eight defs of at most 32 byte classes each); the values equal the interleaved layout's. */
int hrx_witness_batch_device_planes(hrx_ctx *ctx, int layout, const uint8_t *chars, size_t stride, const uint32_t *lens, size_t B, size_t M,
                                    uint32_t *const *record_planes, size_t n_planes, uint16_t *masked, uint64_t *status, void *stream);
void hrx_position_major_plane_sizes(size_t B, size_t M, size_t *plane_u32, size_t *masked_u16);
void hrx_position_major_stripe_sizes(size_t B, size_t M, size_t n_stripes, size_t *stripe_u32, size_t *masked_u16);
/* The n_planes record buffers (the D planes; one def: 1, or 2 row stripes) and the masked rows of a batch of B strings x M rows, each allocated on ctx's device in a
 * neighbourhood of its own: a pool of n_planes + 4 record-sized and 5 masked-row-sized candidates, allocated one after the other (they walk down the device memory), is
 * measured pair by pair (two equal write streams, ~1 ms per pair on the device clock); pairings fall into two levels — both buffers in one class of the physical address
 * space, or not — and the buffers whose busiest class takes the fewest of the launch's output bytes per row (4 per plane — 2 per row stripe —, 2 for the masked rows; then
 * the fewest colliding pairings, the largest sum of pairings) are kept; a pool whose best set still collides (up to three buffers: at all; more: beyond one record buffer +
 * the masked rows) grows by three record candidates, at most twice; then the context's own launch of B strings x M rows runs over the (at most six) best sets by that
 * score, on a constant input, and the fastest is kept (the pairwise probe does not see everything: 2.5 ms against 2.9 per launch for sets it ranks alike); everything else
 * is freed before the call returns.  Record buffers below 1 GiB are carved out of up to
 * four 2-GiB STRIPE ARENAS per device and process chosen the same way once (a probe over buffers that fit the Infinity Cache would measure the cache); hrx_device_free gives
 * a sub-buffer's range back.  hrx_alloc_last_report: steps = pairings measured, ref_gbs = the slowest pairing seen, first_gbs = the slowest pairing of the first buffers
 * (what plain allocations would have been), best_gbs = the kept set's, chosen_step = the bytes per row of the kept set's busiest class, searched = 2: served from arenas
 * measured earlier.  The pool never takes more than 70 % of the free memory (hrx_ctx_set_placement narrows that and bounds the time).  Less than 128 MiB of records,
 * HRX_PLACE_OFF: plain allocations.  record_planes: n_planes pointers out; each buffer is released with hrx_device_free.  n_planes = 1: hrx_alloc_outputs_position_major.
 * Takes the context's lock; not inside a stream capture. */
int hrx_alloc_output_planes(hrx_ctx *ctx, size_t B, size_t M, size_t n_planes, uint32_t **record_planes, uint16_t **masked);
/* The same for a batch that already lies on the device: the launches that choose among the best sets run THE CALLER'S batch (chars, stride, lens, layout as
 * hrx_witness_batch_device_planes takes them: HRX_LAYOUT_POSITION_MAJOR, optionally | HRX_LAYOUT_INPUT_POSITION_MAJOR) instead of a constant stand-in, so where the input lies
 * and what the walk does with it are part of what is measured — the choice for a prover that keeps its input buffer and launches batch after batch into the outputs.  With it
 * one def's single buffer (n_planes = 1) also goes through the pool (on a stand-in input that lost against the pair walk: with one def the input is a seventh of the traffic,
 * profiles/r06_probes/cfg5_pool_dry.txt).  chars / lens must be complete on the device before the call (the launches run on the context's stream); the outputs they write are
 * the candidates', the batch itself is only read. */
int hrx_alloc_output_planes_for_batch(hrx_ctx *ctx, int layout, const uint8_t *chars, size_t stride, const uint32_t *lens, size_t B, size_t M, size_t n_planes,
                                      uint32_t **record_planes, uint16_t **masked);
/* The allocator's measurement, for a caller that manages device memory itself (a prover with its own pool): two equal, time-aligned write streams over the first `bytes`
 * of device buffers a and b (both are OVERWRITTEN), *gbs = bytes written per time.  Pairings in one class of the physical address space measure 5.2-6.2 TB/s, in different
 * classes 6.5-7.3 (a level per box: compare pairings with each other, not with a constant).  Synchronous; takes the context's lock. */
int hrx_probe_write_pair(hrx_ctx *ctx, void *a, void *b, size_t bytes, double *gbs);
/* hrx_rows_of_string_position_major for record planes (or the two row stripes of one def) in HOST memory, each buffer copied from the device as it is: records [M][D], masked [M] of string b. */
int hrx_rows_of_string_planes(const uint32_t *const *record_planes, size_t n_planes, const uint16_t *masked_pm, size_t B, size_t M, size_t D, size_t b,
                              uint32_t *records, uint16_t *masked);
/* One circuit's view of a position-major batch, on the HOST: the rows of string b of a batch of B strings x M rows x D defs that lies in host memory in
 * HRX_LAYOUT_POSITION_MAJOR (e.g. copied from the device as it is) -> records [M][D] u32 and masked [M] u16, the string-major values the fill loops of
 * match_substrs index per string (src/lib.rs:387-519).  A string's consecutive quads are nb*16*D bytes apart, so this is M/4*D + M/8 gathers of 16 bytes:
 * ~10 us per 1024-row string on one core (bench.py: end_to_end_host.gather_us_per_string) against the ~(14 D + 21) gate calls per row that consume
 * them.  Either buffer pair may be NULL (records_pm with records, masked_pm with masked).  No context, no device, re-entrant. */
int hrx_rows_of_string_position_major(const uint32_t *records_pm, const uint16_t *masked_pm, size_t B, size_t M, size_t D, size_t b,
                                      uint32_t *records, uint16_t *masked);
/* Allocates the records and masked-row buffers of a position-major batch of B strings x M rows (sizes as
 * hrx_position_major_sizes) on ctx's device; each is released with hrx_device_free.  Optional — every entry point takes any
 * device pointer — and for records below 128 MiB (a launch that lives in the 256-MB Infinity Cache) just two hipMalloc calls.
 * From there on the call is PLACEMENT-AWARE: two concurrent write streams run ~15 % slower on an MI355X when they lie in the
 * same class of the physical address space (four classes, chosen by address bits >= 2^33) than when they do not, a launch
 * writes records and masked rows as two such streams, and buffers allocated one after the other come from one neighbourhood.
 * So the call walks down the device memory — block after block, each measured against the records with a two-stream write of
 * a few hundred microseconds that times itself on the device clock (robust under a profiler); the reference is the same probe
 * inside ONE block, a candidate that writes 10 % more bytes per time than that is taken, failing that the fastest measured, and everything else is freed
 * before the call returns; the walk never holds more than 70 % of the device memory that was free.
 *   records >= 1 GiB: the blocks are the masked-row candidates themselves, measured against the records buffer itself.
 *   records <  1 GiB: such buffers are carved out of two 2-GiB ARENAS per DEVICE and process (one for records, one for masked
 *     rows; a probe over buffers that fit the Infinity Cache would measure the cache, and the driver puts small blocks into any
 *     hole): the first such call of any context of the device finds the pair, later calls — of every context of that device:
 *     one context per worker thread shares one pair — are served from it until it is full (then a new pair is found);
 *     hrx_device_free returns a sub-buffer's range to its arena (it is handed out again: allocating and freeing per batch stays
 *     on one measured pair) and an arena is released with its last sub-buffer once the pair has been replaced or the device's
 *     last context is destroyed.  A process therefore holds up to 4 GiB per device for its small
 *     output buffers, however many contexts it keeps.
 * 262144 x 2048 B at D = 2 runs at 0.97-1.03 ms with such a pair against 1.12-1.19 ms in a fresh process with two plain
 * allocations (csrc/hrx_place.hip).  The call takes the context's lock, launches on the context's own stream
 * and waits for it: not inside a stream capture.  Per context: hrx_ctx_set_placement (off / on, memory and time budget of a walk).
 * Environment, read by hrx_ctx_create (the defaults of a new context): HRX_PLACE=0 (plain allocations), HRX_PLACE_MAX_STEPS,
 * HRX_PLACE_TRACE=1.  The reference has no counterpart: its witness lives in host Vecs. */
int hrx_alloc_outputs_position_major(hrx_ctx *ctx, size_t B, size_t M, uint32_t **records, uint16_t **masked);
/* The same for any pair of output buffers given in bytes (string-major outputs: B * rec_pitch * D * 4 and B * msk_pitch * 2). */
int hrx_alloc_output_pair(hrx_ctx *ctx, size_t records_bytes, size_t masked_bytes, void **records, void **masked);
/* What the context's last hrx_alloc_output_pair / hrx_alloc_outputs_position_major call did. */
typedef struct hrx_place_report {
    int searched;                /* 0: two plain allocations (small buffers, HRX_PLACE=0, or no memory to walk with); 1: this call walked;
                                    2: served from the arena pair an earlier call measured (the numbers below are that walk's) */
    int steps;                   /* candidates measured */
    int accepted;                /* 1: the kept candidate is >= 10 % above both the same-neighbourhood reference and the MEDIAN candidate of the walk (at
                                    least four candidates; eight in a context's first direct walk) and within 4 % of the best pairing earlier walks of the same kind measured — or a walk that ran
                                    into a bound with a candidate >= 10 % above the reference in hand; 0: simply the fastest measured (csrc/hrx_place_rule.hpp) */
    int chosen_step;
    double ref_us;               /* the reference: both probe streams inside one neighbourhood (device clock) */
    double first_us, best_us;    /* the first candidate (what two plain allocations would have been) and the kept one */
    double ref_gbs, first_gbs, best_gbs;   /* the same three as bytes written per time (GB/s): what acceptance compares — the reference pass
                                    writes fewer bytes than a candidate pass */
    size_t probe_bytes;          /* bytes one probe pass writes */
    size_t peak_candidate_bytes; /* most memory the walk held at once: rejected candidates stay allocated, as the spacers that push the
                                    next candidate further, until the walk ends */
    double search_ms;            /* host time of the whole call */
    int capped;                  /* which bound ended the walk, if one did (0: the acceptance rule itself): HRX_PLACE_CAPPED_* or-ed */
} hrx_place_report;
enum {
    HRX_PLACE_CAPPED_STEPS = 1,  /* the candidate count (48; 2-GiB arena candidates 96; HRX_PLACE_MAX_STEPS) */
    HRX_PLACE_CAPPED_BYTES = 2,  /* the memory budget: 70 % of the free memory, hrx_ctx_set_placement's max_bytes, or — arena walks past their 24th
                                    candidate — less than 30 % of the device left free */
    HRX_PLACE_CAPPED_TIME = 4,   /* hrx_ctx_set_placement's max_ms, or the rule's own hard bound (2 s; arena walks 8 s) */
    HRX_PLACE_CAPPED_ALLOC = 8   /* a candidate could not be allocated */
};
int hrx_alloc_last_report(const hrx_ctx *ctx, hrx_place_report *out);
/* The same into a buffer of out_bytes: min(out_bytes, sizeof(hrx_place_report)) bytes are written.  hrx_place_report only ever grows at its end (`capped` came in round 5), so a
 * consumer built against an earlier header passes the size of ITS struct and gets the fields it knows (hrx_alloc_last_report writes the whole current struct). */
int hrx_alloc_last_report_sized(const hrx_ctx *ctx, void *out, size_t out_bytes);
/* Placement per context (a prover that links the library decides per context, not through the environment):
 *   mode       HRX_PLACE_OFF: hrx_alloc_output_pair / hrx_alloc_outputs_position_major are two plain allocations, nothing is measured, no arena is held
 *              on this context's behalf; HRX_PLACE_WALK (the default unless HRX_PLACE=0 was set when the context was created): as described above.
 *   max_bytes  the most device memory one walk may hold at a time in candidates (0: 70 % of what is free when the walk begins; never more than that)
 *   max_ms     wall-clock bound of one walk in milliseconds (0: the rule's own bounds); the walk keeps the fastest candidate measured until then
 * A walk that ends on a bound says so in hrx_place_report.capped.  Applies to later calls; buffers already handed out are not touched. */
enum { HRX_PLACE_OFF = 0, HRX_PLACE_WALK = 1 };
int hrx_ctx_set_placement(hrx_ctx *ctx, int mode, size_t max_bytes, double max_ms);
/* Roofline diagnostic: the memory traffic of ONE position-major witness launch of B strings x M rows of this context's
 * config (HRX_LAYOUT_POSITION_MAJOR | HRX_LAYOUT_INPUT_POSITION_MAJOR) and nothing else — the input read in 16-byte chunks
 * per lane, every record plane and the masked rows written at the launch's own addresses with the launch's store policy, by
 * four reader and four writer waves per CU — no DFA work.  OVERWRITES records and masked with junk.  What it takes is this
 * box's ceiling for the launch's byte mix on these very buffers (bench.py: roofline.mix_ceiling).  Asynchronous on `stream`. */
int hrx_traffic_pass_device(hrx_ctx *ctx, const uint8_t *chars, size_t stride, size_t B, size_t M, uint32_t *records,
                            uint16_t *masked, void *stream);
/* The same diagnostic for either layout: HRX_LAYOUT_POSITION_MAJOR | HRX_LAYOUT_INPUT_POSITION_MAJOR is the call above (pitches ignored);
 * HRX_LAYOUT_STRING_MAJOR moves the bytes of a string-major launch (M % 8 == 0) the way the walker/storer kernel does — a store instruction writes the 128-byte
 * lines of eight strings, 2 D lines of records and one of masked rows per string and 64 rows, the input read one string per lane — rec_pitch / msk_pitch in rows
 * (0 = M).  What the string-major lines of a sweep are measured against (the reference's fill loops index one string: src/lib.rs:387-519). */
int hrx_traffic_pass_device_layout(hrx_ctx *ctx, int layout, const uint8_t *chars, size_t stride, size_t B, size_t M, uint32_t *records, size_t rec_pitch,
                                   uint16_t *masked, size_t msk_pitch, void *stream);
/* ... over record planes (hrx_witness_batch_device_planes): the same traffic with plane d written at record_planes[d]. */
int hrx_traffic_pass_device_planes(hrx_ctx *ctx, const uint8_t *chars, size_t stride, size_t B, size_t M, uint32_t *const *record_planes, size_t n_planes,
                                   uint16_t *masked, void *stream);
/* From the reference's input shape to the coalesced one, on the device: `chars` = B strings of `stride` bytes each, back to back — one contiguous
 * &[u8] per string is what RegexVerifyConfig::match_substrs is handed (src/lib.rs:311-315) — -> `chars_pm` (B * stride bytes, another buffer) in
 * HRX_LAYOUT_INPUT_POSITION_MAJOR.  Pure streaming (2 * stride bytes of traffic per string; measured beside the bench line: bench.py
 * roofline.from_string_major_input).  Asynchronous on `stream`; no context state is touched. */
int hrx_chars_to_position_major_device(hrx_ctx *ctx, const uint8_t *chars, size_t stride, size_t B, uint8_t *chars_pm, void *stream);
/* Releases a buffer of hrx_alloc_output_pair / hrx_alloc_outputs_position_major / hrx_alloc_output_planes (NULL: nothing).  As with hipFree, the memory is not
 * handed out again before the buffer's device has finished everything in flight, so a buffer may be freed while the launch that writes it is still running.  A buffer of
 * its own: hipFree (which waits).  A sub-buffer of a shared arena: the call returns at once — from any thread, also while another thread captures a stream — and the range
 * is parked; a later allocation whose request does not fit otherwise waits for the device once and takes the parked ranges back. */
int hrx_device_free(void *ptr);
/* Which kernel and launch geometry the planner picks for a batch of B strings x M rows in `layout` on a gfx950 device
 * with `num_cus` compute units (MI355X: 256), as text: "hrx::witness_pm_kernel<1, false, false, false, false, false> grid=256
 * waves=12 ring=4 lds=..." — the kernel name a profiler will show, every template argument spelled out (bench.py's roofline.kernel).  Host-only: nothing is
 * launched and no device is touched.  Returns HRX_OK, HRX_ERR_BOUNDS if nothing fits. */
int hrx_describe_launch(const hrx_defs *defs, int layout, size_t B, size_t M, int num_cus, char *out, size_t cap);
/* The same for a context: its own device's CU count (256 on a host-only context), the kernel-selection flags it was created with and its hrx_ctx_set_option choices. */
int hrx_ctx_describe_launch(const hrx_ctx *ctx, int layout, size_t B, size_t M, char *out, size_t cap);
/* Same with HOST buffers (any alignment/stride >= max len): staged through ctx-owned device buffers, or — below the
 * context's host threshold, and always on a host-only context — walked on the host; synchronous.  This is what an unmodified caller of
 * match_substrs' seam gets (host Vecs in, host Vecs out, src/lib.rs:311-318).  Batches of more than ~100 MiB of rows go one of two ways: pipelined chunk by chunk
 * (a staging thread copies chunk c in and launches its walk while the calling thread copies chunk c - 1's rows out on a second stream), or in, walk, out on one stream —
 * the context times both over its first calls and keeps the faster one for this process on this box (the pipeline's copies out run at half rate on some hosts), looking again
 * every 64 calls; the call lasts about as long as the copy out — (4 D + 2) bytes per row over the PCIe link in one direction (bench.py: end_to_end_host).
 * Environment: HRX_HOST_PIPELINE=1 / 0 forces a way, HRX_HOST_TRACE=1 prints a line per call on stderr, HRX_HOST_CHUNK_MIB the pipeline's chunk size. */
int hrx_witness_batch_host(hrx_ctx *ctx, const uint8_t *chars, size_t stride, const uint32_t *lens, size_t B,
                           size_t M, uint32_t *records, uint16_t *masked, uint64_t *status);

/* ------------------------------------------------------------------ */
/* SURVEY §8 f4 — compact witness -> field cells (the step after the path)        */
/* ------------------------------------------------------------------ */
/* Expands the compact rows of strings [b_begin, b_begin + b_count) of a finished batch into what
 * `Value::known(F::from(v))` holds for every advice cell the reference assigns (src/lib.rs:339-418, 473-519) and for the
 * two result columns of AssignedRegexResult (lib.rs:752-771), with F = halo2curves bn256::Fr, the field the reference's
 * circuits use (lib.rs:896): 4 little-endian u64 limbs per cell in Montgomery form (v * 2^256 mod r — the in-memory
 * representation of Fr, so that a device-side prover or an `assign_advice` loop can take the cells as they are);
 * HRX_FR_CANONICAL in `flags`: the plain integer instead.
 * cells: device buffer [n_cols][b_count][M][4] u64, n_cols = hrx_fr_num_columns(D) = 4 + 4 D, in column order
 *   0 char_enable, 1 characters, 2+4d states[d], 3+4d substr_ids[d], 4+4d start_enable[d], 5+4d end_enable[d],
 *   2+4D masked_characters, 3+4D all_substr_ids.
 * `layout`, chars/stride/lens, records/masked (and rec_pitch/msk_pitch, string-major only; 0 = M) are those of the
 * hrx_witness_batch_device* call that produced the rows.  Any b_count (served 32768 strings per launch).  Asynchronous on `stream`. */
enum { HRX_FR_CANONICAL = 1 };
size_t hrx_fr_num_columns(size_t D);
int hrx_fr_columns_device(hrx_ctx *ctx, int layout, const uint8_t *chars, size_t stride, const uint32_t *lens,
                          const uint32_t *records, size_t rec_pitch, const uint16_t *masked, size_t msk_pitch, size_t B,
                          size_t M, size_t b_begin, size_t b_count, uint64_t *cells, int flags, void *stream);
/* The same out of RECORD PLANES (hrx_witness_batch_device_planes: the D planes, or the two row stripes of one def); `layout` position-major, optionally | HRX_LAYOUT_INPUT_POSITION_MAJOR. */
int hrx_fr_columns_device_planes(hrx_ctx *ctx, int layout, const uint8_t *chars, size_t stride, const uint32_t *lens, const uint32_t *const *record_planes, size_t n_planes,
                                 const uint16_t *masked, size_t B, size_t M, size_t b_begin, size_t b_count, uint64_t *cells, int flags, void *stream);
/* F::from(v) on the host (same arithmetic as the kernel): limbs[4]. */
void hrx_fr_from_u64(uint64_t v, int flags, uint64_t *limbs);

/* ------------------------------------------------------------------ */
/* SURVEY §8 f3 — from the compact records to what the reference's fill consumes (host only, no context, re-entrant)   */
/* ------------------------------------------------------------------ */
/* One circuit: EXACTLY the Vecs the three derive_* calls at the top of match_substrs return (src/lib.rs:316-318), out of the string's compact records, so that the body
 * below lib.rs:318 runs unchanged on them (bindings/rust/hrx.rs WitnessOf is this call):
 *   records    [M][D] u32, string-major rows of ONE string (a row of hrx_witness_batch_host's output, or hrx_rows_of_string_position_major / _planes)
 *   n          characters.len() <= M
 *   states     [D][n + 1] u64   derive_states          (lib.rs:804-823); states[d][n] = the state after the last character
 *   substr_ids [D][n]     usize derive_substr_ids      (lib.rs:825-845)
 *   is_start   [D][n + 1] bool  derive_is_start_end.0  (lib.rs:847-888); is_start[d][n] = false
 *   is_end     [D][n + 1] bool  derive_is_start_end.1;                   is_end[d][0] = false
 * Exact: for i < n the enable cell is 1, so start_enable[i] = is_start[i] and end_enable[i] = is_end[i + 1] (lib.rs:482-519).  When n == M the records have no row n:
 * states[d][M] is returned as 0 and is_end[d][M] as false — the two values the reference computes and never assigns to a cell (lib.rs:388-418, 501).
 * Strings whose status word is not ok have no records to decode (the reference panics in derive_states, lib.rs:817). */
int hrx_witness_of_string(const uint32_t *records, size_t D, size_t n, size_t M, uint64_t *states, size_t *substr_ids, uint8_t *is_start, uint8_t *is_end);
/* All circuits of a batch, column-major: the integer content of every advice cell the loops of match_substrs assign (lib.rs:339-348 char_enable / characters,
 * 419-519 states / substr_ids / start_enable / end_enable per def) and of the two result columns (lib.rs:752-771), for strings [b_begin, b_begin + b_count) of a finished
 * batch that lies in HOST memory in `layout` (string-major with rec_pitch / msk_pitch in rows, 0 = M; or position-major as copied from the device):
 *   columns  [n_cols][b_count][M] u64, n_cols = hrx_witness_num_columns(D) = 4 + 4 D, in the column order of hrx_fr_columns_device:
 *            0 char_enable, 1 characters, 2+4d states[d], 3+4d substr_ids[d], 4+4d start_enable[d], 5+4d end_enable[d], 2+4D masked_characters, 3+4D all_substr_ids
 * — what `Value::known(F::from(v))` is applied to, one contiguous [M] run per circuit and column: a batch fill assigns them without a derive_* call or a gate evaluation per row. */
size_t hrx_witness_num_columns(size_t D);
int hrx_witness_columns_host(int layout, const uint8_t *chars, size_t stride, const uint32_t *lens, const uint32_t *records, size_t rec_pitch, const uint16_t *masked,
                             size_t msk_pitch, size_t B, size_t M, size_t D, size_t b_begin, size_t b_count, uint64_t *columns);

/* ------------------------------------------------------------------ */
/* Multi-GPU driver for host buffers: strings are independent given the RegexDefs, so a batch shards by string index with
 * no collective.  One context per listed device (a device may be listed more than once: its shards then
 * run on separate streams); hrx_multi_witness_batch_host cuts the batch with hrx_shard_range, runs the shards
 * concurrently (one host thread per shard: staging copies and kernels of different devices overlap) and returns when
 * all are done.  Results are identical to one hrx_witness_batch_host call over the whole batch.  On error the first
 * failing shard's status is returned and hrx_last_error() tells which. */
typedef struct hrx_multi hrx_multi;
int hrx_multi_create(const hrx_defs *defs, const int *devices, int n_devices, hrx_multi **out);
void hrx_multi_destroy(hrx_multi *m);
int hrx_multi_num_shards(const hrx_multi *m);
int hrx_multi_shard_device(const hrx_multi *m, int shard);
int hrx_multi_witness_batch_host(hrx_multi *m, const uint8_t *chars, size_t stride, const uint32_t *lens, size_t B, size_t M,
                                 uint32_t *records, uint16_t *masked, uint64_t *status);
/* Device-resident shards: the same driver without the PCIe copies, for a single-process host (the Rust prover) whose
 * inputs already sit in each device's HBM.  Shard r = counts[r] strings whose buffers chars[r], lens[r], records[r],
 * masked[r], status[r] are DEVICE pointers on the device of shard r (hrx_multi_shard_device), in `layout` (one of the
 * hrx_witness_batch_device_layout layouts, each shard a complete array of its own strings).  One kernel per shard is
 * enqueued on the shard's own stream — asynchronous; hrx_multi_synchronize waits for all of them.  No collective.
 * ORDERING: the shard streams are the contexts' private non-blocking streams; nothing orders them against the stream that
 * produced a shard's inputs.  The inputs must be complete before the call, or the caller makes the shard stream
 * (hrx_multi_shard_stream, a hipStream_t) wait on an event recorded behind the producer; the outputs are complete after
 * hrx_multi_synchronize (or behind an event recorded on the shard stream). */
void *hrx_multi_shard_stream(const hrx_multi *m, int shard);
int hrx_multi_witness_batch_device(hrx_multi *m, int layout, const uint8_t *const *chars, size_t stride, const uint32_t *const *lens,
                                   const size_t *counts, size_t M, uint32_t *const *records, uint16_t *const *masked,
                                   uint64_t *const *status);
int hrx_multi_synchronize(hrx_multi *m);

/* Contiguous shard [begin, begin+count) of a batch of B strings for `rank` of `world` devices
 * (strings are independent given the RegexDefs; no collective on the path). */
void hrx_shard_range(size_t B, int world, int rank, size_t *begin, size_t *count);

/* ------------------------------------------------------------------ */
/* Reference-shaped single-string entry points — src/lib.rs:804-888     */
/* ------------------------------------------------------------------ */
/* derive_states(&self, characters:&[u8]) -> Vec<Vec<u64>>: states[d*(n+1) + i].  On the reference's panic
 * returns HRX_ERR_INVALID_TRANSITION and hrx_last_error() is exactly
 * "The transition from {state} by {char} is invalid!" (lib.rs:817). */
int hrx_derive_states(hrx_ctx *ctx, const uint8_t *characters, size_t n, uint64_t *states);
/* derive_substr_ids(&self, states) -> Vec<Vec<usize>>: substr_ids[d*n + i] */
int hrx_derive_substr_ids(hrx_ctx *ctx, const uint64_t *states, size_t n, uint64_t *substr_ids);
/* derive_is_start_end(&self, states, substr_ids) -> (Vec<Vec<bool>>, Vec<Vec<bool>>): each [d*(n+1) + i] */
int hrx_derive_is_start_end(hrx_ctx *ctx, const uint64_t *states, const uint64_t *substr_ids, size_t n,
                            uint8_t *is_start, uint8_t *is_end);
/* match_substrs(&self, ctx, characters) integer columns for one string (lib.rs:311-773); any pointer may be NULL.
 * enable/character/masked_char/masked_substr_id: [M]; state/substr_id/start_enable/end_enable: [D][M]. */
int hrx_match_substrs(hrx_ctx *ctx, const uint8_t *characters, size_t n, size_t M, uint64_t *enable,
                      uint64_t *character, uint64_t *state, uint64_t *substr_id, uint64_t *start_enable,
                      uint64_t *end_enable, uint64_t *masked_char, uint64_t *masked_substr_id,
                      uint64_t *status);

/* ------------------------------------------------------------------ */
/* Definition generation: regex -> minimal DFA (host only, no GPU)     */
/* ------------------------------------------------------------------ */

/* The table step in front of the witness path, without V8: what DecomposedRegexConfig::gen_regex_files
 * (src/vrm/mod.rs:62-95) obtains from get_dfa_json_value -> regexToDfa (src/vrm/js_caller.rs:43-48,
 * src/vrm/regex.js:40-92) and dfa_to_regex_def_text (src/vrm/js_caller.rs:127-157).  `regex` is the concatenation of
 * the parts' regex_def, UTF-8, in the dialect of regex.js:236-367.  Byte-identical to the reference's output
 * (state numbering included).  Two-call pattern: *needed receives the byte length of the result; min(needed, cap)
 * bytes are written to out (no terminator); out may be NULL with cap 0.  A malformed pattern returns HRX_ERR_PARSE
 * with the parser's message ("Error: empty input at 2." ...) in hrx_last_error(). */
int hrx_regex_to_allstr_text(const char *regex, size_t regex_len, char *out, size_t cap, size_t *needed);
/* The intermediate value itself: the JSON text regexToDfa returns ([{"type":..,"edges":{key:target}}, ...]). */
int hrx_regex_to_dfa_json(const char *regex, size_t regex_len, char *out, size_t cap, size_t *needed);

/* DecomposedRegexConfig::gen_regex_files (src/vrm/mod.rs:62-307) as a whole: the AllstrRegexDef text of the
 * concatenated parts and one SubstrRegexDef text per public part (extract_substr_ids, mod.rs:309-538;
 * get_substr_defs_from_path, mod.rs:540-600; text format mod.rs:268-304).  The part regexes are searched in the
 * DFA's paths with leftmost-first semantics (what fancy-regex 0.11 / the regex crate do for these patterns);
 * pattern syntax outside the subset formatRegexPrintable emits from the DFA dialect -> HRX_ERR_PARSE. */
typedef struct hrx_regex_part {
    const char *regex_def; /* UTF-8, not terminated */
    size_t regex_len;
    int is_public;
    size_t max_size;
} hrx_regex_part;
typedef struct hrx_regex_files hrx_regex_files;
int hrx_gen_regex_files(const hrx_regex_part *parts, size_t n_parts, size_t max_byte_size, hrx_regex_files **out);
size_t hrx_regex_files_num_substrs(const hrx_regex_files *files);
/* pointers stay valid until hrx_regex_files_destroy; texts are not terminated, *len receives the byte length */
const char *hrx_regex_files_allstr(const hrx_regex_files *files, size_t *len);
const char *hrx_regex_files_substr(const hrx_regex_files *files, size_t idx, size_t *len);
void hrx_regex_files_destroy(hrx_regex_files *files);
/* format_regex_str (src/vrm/js_caller.rs:36-41 -> formatRegexPrintable, src/vrm/regex.js:24-39); two-call pattern */
int hrx_format_regex_str(const char *regex, size_t regex_len, char *out, size_t cap, size_t *needed);
/* The search gen_regex_files runs per part and path: leftmost-first match of `pattern` in `text`.
 * *found = 0/1; [*start, *end) the match. */
int hrx_regex_find(const char *pattern, size_t pattern_len, const char *text, size_t text_len, int *found, size_t *start,
                   size_t *end);

#ifdef __cplusplus
}
#endif
#endif /* HRX_H */
