// hrx_kernel.hpp — host-visible launch interface of the HIP kernels (hrx_kernel.hip: planner, dispatch, auxiliary kernels; hrx_kernel_pm.hip / hrx_kernel_sm.hip: the witness kernels).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

#include "hrx_defs.hpp"

namespace hrx {

struct LaunchInfo;
struct WitnessArgs;
constexpr uint32_t kMaxDefsPerLaunch = 8;   // RegexDefs per config: the status word carries 8 accept bits (include/hrx.h)

struct WitnessArgs {
    const uint8_t *chars;
    uint64_t stride;
    const uint32_t *lens;
    uint32_t B, M;
    uint32_t rec_pitch, msk_pitch;  // rows between consecutive strings in records / masked (>= M), string-major layout
    uint32_t layout;                // 0 string-major [B][pitch][D] / [B][pitch]; 1 position-major [M/4][D][B][4] / [M/8][B][8]
    uint32_t *records;
    uint16_t *masked;
    uint64_t *status;
    const uint32_t *table_image;  // device copy of DefsSet::table_image
    uint32_t table_bytes;
    const uint64_t *wide_image;   // device copy of DefsSet::wide_image (same byte size as table_image), or NULL
    const uint16_t *half_image;   // device copy of DefsSet::half_image (the exact LDS image, half_bytes long), or NULL
    uint32_t half_bytes;
    const uint8_t *byte_image;    // device copy of DefsSet::byte.image (the exact LDS image: next-state bytes, disp, pair slots), or NULL
    uint32_t byte_bytes, byte_ptab_off, byte_mul_a4, byte_mul_b4, byte_slot_mask4, byte_dead;   // (mul_a * 4, mul_b * 4, (slots - 1) * 4: byte offsets of the 4-byte pair slots)
    uint32_t byte_one_id;         // BYTE table, position-major: the def's ONE substring id (0: it has several, or none): the finisher then holds three tiles instead of two (hrx_kernel_pm.hip)
    uint32_t byte_rows_bytes, byte16_bytes, byte16_ptab_off;   // the walker/storer kernel's image: next-state bytes [0, byte_rows_bytes) + the 2-byte slots (byte_image + byte_bytes) at byte16_ptab_off
    const uint8_t *cw_image;      // device copy of DefsSet::cw_image (CLASS-WIDE tables of a whole config of 4 .. 8 defs: hrx_kernel_pmd.hip CW; table_bytes = its size then), or NULL
    uint32_t cw_lut_off;          // LDS offset of def 0's class LUT
    const uint8_t *pair_image;    // device copy of DefsSet::pair.image (the exact LDS image: blocks + class LUT), or NULL
    uint32_t pair_bytes, pair_classes, pair_blk_bytes, pair_lut_off;
    uint32_t n_groups;            // ceil(B / gs), set by plan_witness_launch
    uint32_t gs;                  // strings per wave (64, 32 or 16), set by plan_witness_launch
    uint32_t D;
    uint32_t debug;               // the context's kDbg* bits below (forced kernel / table choices for the tests; ablations only with -DHRX_ABLATION)
    uint32_t tune;                // the context's kTune* bits (hrx_ctx_set_option, include/hrx.h): choices between variants that compute the same rows
    unsigned long long *stamps;   // profiling only (tools/kbench): per wave and tile 4 s_memtime stamps; NULL in the product
    // dynamic group assignment (position-major kernel, batches of >= 8 long groups per walker pair): after its first group a pair
    // takes the next unclaimed group from a device counter (hrx_kernel_pm.hip); 0 / NULL: groups g_first + j * stride
    uint32_t *group_counter;
    uint32_t group_base, group_first_dyn;
    // one pass of a multi-pass config (more than kMaxDefsPerPass defs; position-major outputs): this launch walks defs
    // rec_d0 .. rec_d0 + D - 1 of a config of rec_D defs and writes their record planes straight into the caller's buffer
    // ([M/4][rec_D][nb][4]); instead of (meaningless per-group) masked rows its finisher writes the tile summaries the combine
    // kernel needs — per tile and string 80 bytes: ST / EN bitvectors + one substr-id byte per row, [tile][5][B][16 B]
    uint32_t rec_D, rec_d0;       // 0 / 0: this launch's defs are the whole config
    uint32_t *summary;            // NULL: an ordinary launch
    // RECORD PLANES (include/hrx.h hrx_witness_batch_device_planes): def d's records in a buffer of its own — per block of kPmBlock strings [ceil(M/4)][nb][4], the
    // D = 1 position-major layout — instead of plane d of the interleaved [M/4][D][nb][4].  A D = 3 launch writes 12 of its 14 output bytes per row into the records;
    // as ONE allocation they lie in one class of the physical address space and the launch runs at what one class absorbs (the no-compute pass of cfg 4:
    // 0.70-0.77 of peak), as separately placed planes at 0.86 (profiles/r06_probes/plane_probe.txt).  rec_planes[0] == NULL: the interleaved layout, `records`.
    unsigned char *rec_planes[kMaxDefsPerLaunch];
    // ... and ONE def's plane in two ROW STRIPES (rec_stripes == 2, D == 1): quad q of a string in buffer q % 2 at slot q / 2 — per block [ceil(ceil(M/4)/2)][nb][4] each — so that even a
    // one-def launch's 4 record bytes per row spread over two classes of the address space beside the masked rows' third (the no-compute pass of the bench line 0.76 -> 0.805 of peak,
    // of cfg 5's byte count 0.72-0.80 -> 0.82: profiles/r06_probes/stripe_probe_*.txt).  0 / 1: whole planes.
    uint32_t rec_stripes;
    // the LAST pass of a multi-pass config reads the earlier groups' tile summaries itself (its finisher has everything else in hand: this
    // group's bitvectors and id bytes, the input bytes, the reveal-mask carries) and writes the FINAL masked rows: no combine launch,
    // this group's summary never written, the others' read once.  0: not such a pass.
    uint32_t merge_G;                              // earlier groups (<= kMaxMergeGroups)
    const uint32_t *merge_summary[3];              // their WitnessArgs::summary buffers
    uint32_t *merge_ov;                            // [B]: lowest row that defs of DIFFERENT groups flag together (0xffffffff: none) -> witness_merge_status_kernel
    // chunked launch of the loader / walker / finisher kernel (hrx_kernel_spec.hip): NULL / 0 in an ordinary launch
    const uint32_t *vs_init;      // [chunk][B][D]: state | substr id of the transition into the chunk's first row << 16 | its end flag << 24
    uint32_t vs_chunks, vs_tiles, vs_groups;   // chunks per string, tiles per chunk, REAL groups (n_groups = vs_groups * vs_chunks)
    uint64_t *vs_status;          // [chunk][B]: every chunk's status word
    uint2 *vs_info;               // [chunk][B]: pend | fwd << 1 | sm << 2 | dec << 3, pend_start (hrx_lane.h TileMasks)
    uint32_t sm_no_touch;         // walker/storer kernel: 1 = the storer does not warm L2 for the walker (set by plan_witness_launch; hrx_kernel_sm.hip)
    uint32_t sm_bufs;             // def-parallel kernel, string-major outputs: LDS sub-tile buffers per group, 2 or 3 (set by plan_pmd_cw_sm; hrx_kernel_pmd.hip SMO)
    uint32_t nt_mix;              // position-major kernels: which stores are write-back instead of streaming (kNtMix*, hrx_kernel_pm.hip)
    uint32_t pace_even;           // profiling only (HRX_PACE, stamps / ablation builds): x 64 idle cycles per tile for the walkers of even workgroups; 0 in the product
    DefConsts dc[kMaxDefsPerLaunch];
};

// HRX_DEBUG_FLAGS (environment; read ONCE per context in hrx_ctx_create, never per launch).  The "force" bits let the
// parity tests drive every kernel / table format through the same batches; they change which kernel runs, never what it
// computes.  The ablation bits (stores skipped, input re-read from L2, ...) make the OUTPUT WRONG: they exist only in a
// library built with -DHRX_ABLATION (`make ablation` -> libhrx_ablation.so, used by tools/ only); in the release
// libhrx.so their enumerators are 0, so every `a.debug & kDbgSkip...` test folds to false at compile time and no
// environment variable can make a launch skip work.
#ifdef HRX_ABLATION
#define HRX_ABL(x) (x)
#else
#define HRX_ABL(x) 0u
#endif
enum : uint32_t {
    kDbgSkipRecords = HRX_ABL(1u),             // ablation: no record stores
    kDbgSkipMasked = HRX_ABL(2u),              // ablation: no masked-row stores
    kDbgInputFromL2 = HRX_ABL(4u),             // ablation: every tile re-reads the first input lines (no HBM reads) / no prefetch of the next tile
    kDbgNoTouch = HRX_ABL(8u),                 // ablation, walker/storer kernel: no L2 touch-ahead of the input
    kDbgSplitNoWalk = HRX_ABL(16u),            // ablation, walker/storer kernel: storers move whatever the slots hold
    kDbgNoNtRecords = HRX_ABL(32u),            // ablation, position-major kernels: ordinary record stores
    kDbgNoNtMasked = HRX_ABL(64u),             // ablation, position-major kernels: ordinary masked-row stores
    kDbgNoNtStores = HRX_ABL(96u),             // both (the release build's stores are non-temporal at compile time: no branch per store)
    kDbgPpNoTranslate = HRX_ABL(0x100u),      // ablation, pair-step kernel: the loader skips the class lookups (pair index 0 everywhere)
    kDbgPpNoPost = HRX_ABL(0x200u),            // ablation, pair-step kernel: the walker only follows the chain (no records, flags, ids)
    kDbgPpNoMask = HRX_ABL(0x400u),            // ablation, pair-step kernel: no reveal-mask work at the tile end
    kDbgSkipFixups = HRX_ABL(0x800000u),       // ablation: no end-mask fix-ups
    kDbgFixedLines = HRX_ABL(0x1000000u),      // ablation: every quad / octet of a string is stored onto its first one
    kDbgFixToDummy = HRX_ABL(0x4000u),         // ablation: the end-mask repairs are stored onto the string's first octets (same instructions, no read-modify-write at the memory)
    kDbgForceOneWave = 0x10000u,      // string-major: the one-wave kernel instead of the walker/storer kernel
    kDbgGroups32 = 0x20000u,          // one-wave kernel: 32 strings per wave
    kDbgForceGlobalTable = 0x40000u,  // walk the fused table out of global memory even if it fits LDS
    kDbgForceNarrow = 0x80000u,       // position-major kernel: 4-byte table even where the planner picks WIDE
    kDbgForceWide = 0x200000u,        // position-major kernel: WIDE table also at D = 1
    kDbgForceHalf = 0x400000u,        // position-major kernel: HALF table even if the 4-byte one fits LDS
    kDbgForceByte = 0x2000u,          // position-major kernel, one def: BYTE table even if the 4-byte one fits LDS
    kDbgNoByte = 0x8000u,             // position-major kernel: never the BYTE table (the HALF table where the 4-byte one does not fit)
    kDbgNoDefParallel = 0x2000000u,   // position-major: never the def-parallel kernel
    kDbgForceDefParallel = 0x4000000u,// position-major, D >= 2, WIDE table: the def-parallel kernel whatever the batch size (tests)
    kDbgNoPair = 0x8000000u,          // position-major, D = 1: never the pair-step kernel (hrx_kernel_pp.hip)
    kDbgForcePair = 0x40000000u,      // position-major, D = 1 with a PAIR table: the pair-step kernel whatever the batch size (tests, A/B)
    kDbgNoDynamicGroups = 0x1u << 11,  // position-major kernel: static group assignment whatever the batch size
    kDbgForceDynamicGroups = 0x1u << 12,   // ... dynamic from two groups per walker and any row count (tests)
    kDbgXcdRemap = 0x100000u,         // position-major kernels: every XCD walks a CONTIGUOUS eighth of the groups (hrx_device.h xcd_slot; measured 1.5 % slower, off by default)
    kDbgForceSpec = 0x80u,            // position-major: the chunked launch (hrx_kernel_spec.hip) whatever the batch size, chunks of 4 tiles (tests)
    kDbgNoSpec = 0x80000000u,         // position-major: never the chunked launch
    kDbgForceHost = 0x10000000u,      // host-buffer entry points: always the native host walk (hrx_host_walk.cpp)
    kDbgNoHost = 0x20000000u,         // host-buffer entry points: never the native host walk
    // every bit that merely selects a kernel (the only ones a release build honours)
    kDbgForceMask = kDbgForceOneWave | kDbgGroups32 | kDbgForceGlobalTable | kDbgForceNarrow | kDbgForceWide | kDbgForceHalf | kDbgForceByte | kDbgNoByte |
                    kDbgNoDefParallel | kDbgForceDefParallel | kDbgNoPair | kDbgForcePair | kDbgXcdRemap | kDbgNoDynamicGroups | kDbgForceDynamicGroups | kDbgForceHost | kDbgNoHost |
                    kDbgForceSpec | kDbgNoSpec,
#ifdef HRX_ABLATION
    kDbgHonoured = 0xffffffffu,
#else
    kDbgHonoured = kDbgForceMask,
#endif
};
// hrx_ctx_set_option (include/hrx.h) -> WitnessArgs::tune
enum : uint32_t {
    kTunePmdFinMask = 3u, kTunePmdFinOn = 1u, kTunePmdFinOff = 2u,   // HRX_OPT_PMD_COMBINER_WAVE: the def-parallel kernel on the WIDE table with / without a combiner wave (0: the planner's default)
};
// HRX_DEBUG_FLAGS from the environment, reduced to the bits this build honours
uint32_t debug_flags_from_env();

struct LaunchInfo {
    int split;         // 6: pair-step loader/walker kernel, position-major, D = 1 (witness_pp_kernel: two bytes per dependent lookup),
                       // 5: def-parallel loader/walker kernel for D >= 2 batches that leave walker slots empty (witness_pmd_kernel),
                       // 2: loader/walker kernel for the position-major layout (witness_pm_kernel),
                       // 1: walker/storer kernel (witness_split_kernel), 0: one-wave-does-all kernel (witness_kernel)
    int waves_per_wg;  // split: 2 * pairs
    int nslots;        // split: ring slots per walker/storer pair
    int gtab;          // 1: fused table read from global memory (too large for LDS)
    int wide;          // 1: position-major kernel on the WIDE table (hrx_lane.h)
    int half;          // 1: position-major kernel on the HALF table (hrx_lane.h)
    int byte;          // 1: position-major kernel on the BYTE table (hrx_lane.h): walker + loader + finisher
    int grid;
    int dyn;           // 1: dynamic group assignment (position-major loader/walker kernel, >= 8 long groups per walker pair)
    int spec_tiles;    // > 0: CHUNKED launch (hrx_kernel_spec.hip): tiles per chunk; the geometry above is that of the walk over the chunks
    int spec_chunks;   // chunks per string
    int pmd_fin;       // def-parallel kernel on the WIDE table (two and three defs): 1 = a combiner wave of its own per group (hrx_kernel_pmd.hip FIN)
    size_t lds_bytes;
};

// LDS bytes one wave stages per 64-string x 64-row tile
constexpr size_t wave_stage_bytes(int D, unsigned gs) { return ((size_t)gs + 1) * ((256 * (size_t)D + 16) + 80 + 8); }
constexpr size_t kLdsLimit = 160 * 1024;

// Picks the launch geometry for `a` on a device with `num_cus` CUs; returns false if nothing fits.
bool plan_witness_launch(WitnessArgs &a, int num_cus, LaunchInfo &out);
bool plan_pmd_cw(WitnessArgs &a, int num_cus, LaunchInfo &out);
bool plan_pmd_cw_sm(WitnessArgs &a, int num_cus, LaunchInfo &out);   // ... with string-major outputs straight out of the launch (four and five defs, rows in multiples of 16)   // a whole config of 4 .. 7 defs in one def-parallel launch on the CLASS-WIDE tables (a.cw_image set)
hipError_t launch_witness(const WitnessArgs &a, const LaunchInfo &li, hipStream_t stream);
hipError_t launch_zero_u32(uint32_t *p, hipStream_t stream);   // *p = 0 as a kernel node (hrx_kernel.hip)
// the two kernel translation units behind launch_witness: li.split == 2 -> hrx_kernel_pm.hip, else hrx_kernel_sm.hip
hipError_t launch_witness_pm(const WitnessArgs &a, const LaunchInfo &li, hipStream_t stream);
hipError_t launch_witness_sm(const WitnessArgs &a, const LaunchInfo &li, hipStream_t stream);
hipError_t launch_witness_pmd(const WitnessArgs &a, const LaunchInfo &li, hipStream_t stream);   // hrx_kernel_pmd.hip (split == 5)
hipError_t launch_witness_pp(const WitnessArgs &a, const LaunchInfo &li, hipStream_t stream);    // hrx_kernel_pp.hip (split == 6)
// position-major loader/walker kernel (hrx_kernel_pm.hip): LDS bytes per walker/loader pair.  FIN = the loader also finishes
// the tiles (reveal masks + masked rows) from a 6-KiB summary the walker hands over; not for the HALF table (a 256-state table
// leaves no LDS for it) nor for string-major outputs.
// nt_mix: low byte k: the records of every k-th tile (t % k == k - 1) are stored write-back, 0 = all streaming; bit 8: the masked
// rows write-back.  plan_nt_mix (hrx_kernel.hip) picks k so that ~128 MiB of a launch's records stay write-back.  Whatever it says, the
// masked rows of a tile into which an OPEN optimistic span reaches are written back: they may be zeroed later, and a repair that finds
// its line in L2 costs the memory nothing (hrx_kernel_pm.hip octets_out).  Tools only (the libhrx_ntenv.so build reads HRX_NT_MIX /
// HRX_NT_FLAGS at every launch, tools/ab_policy.py): bit 9 = not even those, bits 12-15 km = the masked rows of every km-th tile as well.
constexpr uint32_t kNtMixMaskedWb = 0x100u, kNtMixNoOpenSpan = 0x200u;
// hrx_place.hip: microseconds (device clock) of a time-aligned two-stream write over two regions (both are overwritten); clk: 16 bytes of device scratch
double placement_probe_us(void *rec, size_t rec_bytes, void *msk, size_t msk_bytes, uint32_t D, hipStream_t st, unsigned long long *clk, size_t *bytes_written);
// hrx_place.hip: the memory traffic of one position-major witness launch of this shape and nothing else (roofline diagnostics)
hipError_t launch_traffic_pass_sm(const uint8_t *chars, size_t stride, size_t B, size_t M, uint32_t D, uint32_t *records, size_t rec_pitch, uint16_t *masked,
                                  size_t msk_pitch, uint32_t nt_mix, uint32_t *sink, int num_cus, hipStream_t stream);
hipError_t launch_traffic_pass(const uint8_t *chars, size_t stride, size_t B, size_t M, uint32_t D, uint32_t *records, uint16_t *masked,
                               uint32_t nt_mix, uint32_t *sink, int num_cus, hipStream_t stream, uint32_t *const *planes = nullptr, uint32_t stripes = 1);
uint32_t plan_nt_mix(const WitnessArgs &a, const LaunchInfo &li);
constexpr size_t kPmSummaryBytes = 6144;
template <bool HALF, bool SM> constexpr bool kPmFinisher = !HALF && !SM;
constexpr int pm_max_threads(bool half, bool sm) { return (!half && !sm) ? 768 : 512; }
constexpr size_t kPmCounterBytes = 128;   // ready / freed / freed2 / sum_ready / sum_freed / gq_ready + the 16-entry group queue
constexpr size_t pm_pair_bytes(size_t nring, bool half, bool fin) { return nring * 4096 + (half ? 0 : 4096) + (fin ? kPmSummaryBytes : 0) + kPmCounterBytes; }
constexpr size_t kPpSlotBytes = 8192;   // pair-step kernel: one ring slot = 4 KiB of pair indices + 4 KiB of raw bytes
constexpr size_t pp_pair_bytes(size_t nring) { return nring * kPpSlotBytes + kPmSummaryBytes + kPmCounterBytes; }   // + the finisher's tile summary + counters
// LDS bytes per group of the def-parallel kernel: input ring + (D - 1) x (2 summaries of 5 KiB + a 2-KiB status piece) + counters
constexpr size_t pmd_group_bytes(int D, int nring) { return (size_t)nring * 4096 + (size_t)(D - 1) * (2 * 5120 + 2048) + 192; }
// ... of the WIDE kernel with a combiner wave (FIN): all D walkers publish, their status pieces lie in their first summary slots (hrx_kernel_pmd.hip PA)
constexpr size_t pmd_fin_group_bytes(int D, int nring) { return (size_t)nring * 4096 + (size_t)D * (2 * 5120) + 192; }

// Configs of more than kMaxDefsPerPass RegexDefs (hrx_defs.hpp): every group of defs is walked by an ordinary launch into a
// group-private position-major records buffer [ceil(M/4)][D_g][nb][4] (blocked like every position-major buffer) and its own
// status array; witness_combine_kernel (hrx_kernel_mp.hip) then reads the groups' rows once, writes them into the caller's
// records buffer in its layout, and computes what needs ALL defs of a row: Sum(substr_id), Sum(is_start), Sum(is_end) ->
// reveal masks and masked rows (lib.rs:467-519, 593-764), the flag-overlap row, the merged status word.
constexpr uint32_t kMaxGroups = 32;
constexpr uint32_t kMaxMergeGroups = 3;   // earlier groups the last pass of a multi-pass config can merge itself (WitnessArgs::merge_summary); more: the combine launch
struct CombineArgs {
    const uint8_t *chars;
    uint64_t stride;
    const uint32_t *lens;
    uint32_t B, M, D, G;
    uint32_t layout;                // the CALLER's layout: bit 0 position-major outputs, bit 1 position-major input
    uint32_t rec_pitch, msk_pitch;  // string-major outputs
    uint32_t *records;
    uint16_t *masked;
    uint64_t *status;
    const uint32_t *grec[kMaxGroups];      // copy mode: the groups' private records buffers; summary mode: NULL
    const uint32_t *gsummary[kMaxGroups];  // summary mode (position-major outputs): the groups' tile summaries (WitnessArgs::summary)
    const uint32_t *merge_ov;              // launch_merge_status: WitnessArgs::merge_ov of the last pass
    const uint64_t *gstatus[kMaxGroups];
    uint8_t gD[kMaxGroups], gfirst[kMaxGroups];
};
hipError_t launch_combine(const CombineArgs &a, hipStream_t stream);
// the status words of a multi-pass config whose last pass merged the summaries itself: one thread per string (bad length, the lowest def's
// undefined transition, the lowest overlap row, the accept bits of all groups)
hipError_t launch_merge_status(const CombineArgs &a, hipStream_t stream);

// CHUNKED walk (hrx_kernel_spec.hip): scout -> compose -> the loader / walker / finisher kernel over chunks (WitnessArgs::vs_*) -> stitch
constexpr uint32_t kSpecMaxChunks = 32, kSpecMaxChunkTiles = 64;
struct SpecArgs {
    const uint8_t *chars;
    uint64_t stride;
    const uint32_t *lens;
    uint32_t B, M, D;
    uint32_t in_pm;                          // 1: position-major input ([stride/16][B][16]), 0: string-major ([B][stride])
    uint32_t C, tiles_per_chunk, n_groups;   // chunks per string, 64-row tiles per chunk, REAL groups of 64 strings
    const uint32_t *table_image;             // the narrow fused table (global copy; the scout stages it into LDS)
    uint32_t table_bytes;
    DefConsts dc[kMaxDefsPerLaunch];
    uint32_t n_states[kMaxDefsPerLaunch];    // real states per def (largest + 1)
    uint32_t qabs[kMaxDefsPerPass][8];       // bit s: real state s is quasi-absorbing — every byte keeps it where it is, except bytes that NO real state
                                             // of the def has a transition for (those send it to the dead row, like everybody else)
    uint32_t smax;                           // >= every n_states
    // COMPACT scout tables (hrx_api.cpp build_scout_image): per def a 256-byte class LUT (byte -> 2 x its column class) and the transition table over
    // CLASSES, u16 entries holding the LDS byte address of the next state's row: a row is a few dozen bytes, so the lanes of a wave — different strings, mostly the
    // same few states — read a handful of dwords instead of 64 random ones (the 4-byte [state][byte] table costs ~4.5 bank-conflict passes per wave read).
    // cimage == NULL: some def has more than 127 classes (or the image passes 64 KiB): the scout walks the narrow table.
    const uint8_t *cimage;
    uint32_t cimage_bytes;
    uint32_t c_lut[kMaxDefsPerPass], c_tab[kMaxDefsPerPass], c_rowb[kMaxDefsPerPass], c_inv[kMaxDefsPerPass];   // LDS offsets of LUT and table, bytes per row, ceil(2^16 / rowb)
    const uint16_t *pair_tags[kMaxDefsPerLaunch];   // device (state, next) -> tag tables (hrx_defs.hpp pair_tags)
    // scratch: one row per (chunk, def, string), [chunk][def][n_groups * 64][row_bytes]: smax bytes "the state start state s reached after
    // the scout's stage A" + a 32-byte record (8 keys | where each key ends | where it is before the chunk's last byte | fail flag)
    uint8_t *rows;
    uint32_t row_bytes;                      // smax + 32, a multiple of 16
    uint32_t dbg;                            // profiling only (ablation build, HRX_SPEC_DBG): 1 no tag load, 2 no init store in compose
    uint32_t *init;                          // [chunk][B][D] -> WitnessArgs::vs_init
    const uint2 *vinfo;                      // [chunk][B] <- WitnessArgs::vs_info
    const uint64_t *vstatus;                 // [chunk][B] <- WitnessArgs::vs_status
    uint64_t *status;
    const uint32_t *records;
    const uint32_t *rec_planes[kMaxDefsPerPass];   // record planes (WitnessArgs::rec_planes): def d's records in a buffer of its own; rec_planes[0] == NULL: the interleaved `records`
    uint16_t *masked;
    // chunks whose reveal-mask assumptions were wrong, found by the stitch launch, recomputed by the repair launch (one wave each)
    uint32_t *work_count;                    // [1], zeroed in front of the stitch launch
    uint2 *work;                             // [work_cap]: string, chunk | true start_mask before it << 8 | true end_mask of its last row << 9
    uint32_t work_cap;
};
hipError_t launch_spec_scout(const SpecArgs &a, int num_cus, hipStream_t stream);
hipError_t launch_spec_compose(const SpecArgs &a, hipStream_t stream);
hipError_t launch_spec_stitch(const SpecArgs &a, int num_cus, hipStream_t stream);

// position-major -> string-major (hrx_kernel_tp.hip): string-major callers served by the position-major path
struct TransposeArgs {
    const uint32_t *records_pm;     // [ceil(M/4)][D][nb][4], blocked by kPmBlock strings
    const uint16_t *masked_pm;      // [ceil(M/8)][nb][8]
    uint32_t B, M, D;
    uint32_t *records;              // [B][rec_pitch][D]
    uint16_t *masked;               // [B][msk_pitch]
    uint32_t rec_pitch, msk_pitch;  // rows, multiples of 8
};
hipError_t launch_transpose(const TransposeArgs &a, hipStream_t stream);
// string-major input bytes [B][stride] -> HRX_LAYOUT_INPUT_POSITION_MAJOR (blocks of kPmBlock strings, [stride/16][nb][16]); hrx_kernel_tp.hip
hipError_t launch_chars_to_position_major(const uint8_t *chars_sm, size_t stride, size_t B, uint8_t *chars_pm, hipStream_t stream);

// states-in entry points (lib.rs:825-888): tags[d*n+i] = pair_tag(states[d][i], states[d][i+1])
hipError_t launch_pair_tags(const uint64_t *states, size_t n, uint32_t D, const uint16_t *const *pair_tags,
                            const uint32_t *n_states, uint16_t *tags, hipStream_t stream);

// derive_is_start_end (lib.rs:847-888) for caller-supplied states AND substr ids:
// flags[d*n+i] bit0 = is_start[d][i], bit1 = is_end[d][i+1]
struct EndpointArgs {
    const uint64_t *states;
    const uint64_t *substr_ids;
    uint64_t n;
    uint32_t D;
    uint8_t *flags;
};
// member[d]: device membership bytes of def d; dims[3 d .. 3 d + 2] = {n_states, n_substrs, id_offset} of def d (one launch per def)
hipError_t launch_endpoint_flags(const EndpointArgs &a, const uint8_t *const *member, const uint32_t *dims, hipStream_t stream);

// compact witness rows of strings [b_begin, b_begin + b_count) -> bn256::Fr cells, [col][string][row][4]
struct FrArgs {
    const uint8_t *chars;
    uint64_t stride;
    const uint32_t *lens;
    const uint32_t *records;
    const uint16_t *masked;
    uint32_t B, M, D, layout, rec_pitch, msk_pitch;
    uint32_t b_begin, b_count;
    uint32_t canonical;   // 1: plain integers instead of Montgomery form
    const uint32_t *rec_planes[kMaxDefsPerLaunch];   // record planes / the two row stripes of one def (WitnessArgs::rec_planes, ::rec_stripes); rec_planes[0] == NULL: `records`
    uint32_t rec_stripes;
    uint64_t *cells;      // first cell of string b_begin in column 0
    uint64_t col_cells;   // cells between consecutive columns (= strings of the whole request x M)
};
hipError_t launch_fr_columns(const FrArgs &a, hipStream_t stream);

}  // namespace hrx
