// hrx_kernel_pm.hip — gfx950 kernel for the POSITION-MAJOR buffers (the headline path).
#include <hip/hip_runtime.h>

#include "hrx_device.h"
#include "hrx_walk_pm.h"

namespace hrx {

// BYTE table (hrx_walk_pm.h walk_tile_pm_byte): the walker hands ONE TAG BYTE per row over — substr id | is_start << 6 | is_end << 7,
// four rows to a dword — and the finisher derives what the reveal-mask scans need from the tile's 16 dwords, eight rows at a time:
// bit 6 / bit 7 of eight tag bytes -> one byte of ST / EN with two v_dot4_u32_u8 (weights 1, 2, 4, 8 and 16, 32, 64, 128; the byte
// arrives shifted by the bit's position), "id differs from the id of the row before" the same way from (ids ^ ids shifted by one row)
// + 0x3f per byte (bit 6 = the byte is not zero; ids are < 64), and the substr-id bytes themselves are the tag bytes & 0x3f.
// ~23 vector instructions per eight rows; one row at a time this cost the walker ~12 per row (lib.rs:831-888 per row).
__device__ __forceinline__ TileBits byte_tile_bits(uint32_t (&sidq)[16], uint32_t &sid_prev) {
    uint32_t st[2] = {0, 0}, en[2] = {0, 0}, ch[2] = {0, 0};
    uint32_t prevw = sid_prev << 24;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        const uint32_t ta = sidq[2 * o], tb = sidq[2 * o + 1];
        const uint32_t sa = ta & 0x3f3f3f3fu, sb = tb & 0x3f3f3f3fu;
        const uint32_t s8 = __builtin_amdgcn_udot4(tb & 0x40404040u, 0x80402010u, __builtin_amdgcn_udot4(ta & 0x40404040u, 0x08040201u, 0u, false), false);   // ST byte << 6
        const uint32_t e8 = __builtin_amdgcn_udot4(tb & 0x80808080u, 0x80402010u, __builtin_amdgcn_udot4(ta & 0x80808080u, 0x08040201u, 0u, false), false);   // EN byte << 7
        const uint32_t ya = __builtin_amdgcn_alignbit(sa, prevw, 24), yb = __builtin_amdgcn_alignbit(sb, sa, 24);                                             // the ids of rows p - 1
        const uint32_t za = ((sa ^ ya) + 0x3f3f3f3fu) & 0x40404040u, zb = ((sb ^ yb) + 0x3f3f3f3fu) & 0x40404040u;
        const uint32_t c8 = __builtin_amdgcn_udot4(zb, 0x80402010u, __builtin_amdgcn_udot4(za, 0x08040201u, 0u, false), false);                               // CH byte << 6
        prevw = sb;
        const int sh = 8 * (o & 3);
        st[o >> 2] |= (s8 >> 6) << sh;
        en[o >> 2] |= (e8 >> 7) << sh;
        ch[o >> 2] |= (c8 >> 6) << sh;
        sidq[2 * o] = sa; sidq[2 * o + 1] = sb;
    }
    sid_prev = prevw >> 24;
    TileBits out;
    out.st = (uint64_t)st[0] | ((uint64_t)st[1] << 32);
    out.en1 = (uint64_t)en[0] | ((uint64_t)en[1] << 32);
    out.ch = (uint64_t)ch[0] | ((uint64_t)ch[1] << 32);
    return out;
}

// =============================================================================================
// Position-major kernel (layout 1): records [ceil(M/4)][D][B][4] u32, masked [ceil(M/8)][B][8] u16.
//
// With one lane per string, four consecutive rows of a lane are 16*D contiguous bytes and the 64 lanes of a wave are
// 64 consecutive strings: every store is a full, contiguous 1-KiB (D=1) run written straight from the walker's
// registers — no LDS transpose, no mover wave — and at any moment the whole chip writes into one compact slab of the
// output (rows 4q..4q+3 of all strings = 1 MiB at B = 65536).  A compact write window is what the HBM write path
// rewards: 6.5 TB/s vs 4.3-5.2 TB/s for the string-major comb (tools/fillprobe, tools/wpattern2; NOTES_MEASUREMENTS.md §4).
//
// The walker's in-order vmcnt would make any wait for an input load also wait for every store issued before it, so
// the walker issues no loads at all: a LOADER wave per walker streams the strings' bytes through its own registers (RT tiles
// of 16 B per lane in flight, counted s_waitcnt) into an LDS ring and the walker picks its 64 bytes per tile up with four
// ds_read_b128.
//
// FIN (every variant but HALF and the string-major one): a tile FINISHER wave.  In-kernel stamps
// (tools/kbench, profiles/r02_probes/stamps_*.txt) show the walker pacing the launch: per 64-row tile ~5500 cycles of
// dependent chain + hidden VALU with all stores skipped, ~2200 more with the stores in (the same for 16 and for 24 store
// instructions per tile: the wave is held by the write path's back-pressure), and ~1450 cycles of tile-end work (mask scans,
// masked-row assembly) that is serial to the chain.  Everything per tile that does not feed the chain
// therefore moves to a third wave per pair, the FINISHER: the walker hands the tile's three bitvectors, its substr-id bytes
// and the string's length over through LDS (96 B per lane, six ds_write_b128) and the finisher — which reads the tile's raw
// bytes from the input ring slot (the loader re-uses a slot only after walker AND finisher are done with it) — runs the
// reveal-mask scans (lib.rs:598-764), the fix-ups and the eight masked-row stores.  It issues no loads, so nothing it does
// disturbs the loader's counted vmcnt waits.  The walker keeps the chain, the records and their stores, the error paths and
// the status word.  Three waves per SIMD: 168 VGPRs each (the loader runs 8 instead of 12 tiles ahead).
// =============================================================================================
template <int D, bool GTAB, bool WIDE, bool HALF = false, bool SM = false, bool BYTE = false>
__global__ __launch_bounds__(pm_max_threads(HALF, SM)) void witness_pm_kernel(const WitnessArgs a, const uint32_t nring) {
    static_assert(!BYTE || (D == 1 && !GTAB && !WIDE && !HALF && !SM), "the BYTE table serves one def, position-major outputs");
    constexpr bool FIN = kPmFinisher<HALF, SM>;
#ifdef HRX_STAMPS
    const unsigned long long wall_entry = wall_clock64();   // 100 MHz, the same clock on every CU
#endif
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t pairs = (blockDim.x >> 6) / (FIN ? 3u : 2u);  // walker waves 0..pairs-1, loader waves pairs..2*pairs-1, FIN: finisher waves 2*pairs..3*pairs-1
    const bool is_walker = wave < pairs, is_finisher = FIN && wave >= 2u * pairs;
    const uint32_t pair = wave % pairs;
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();

    // ring + the walker's 4-KiB scratch (HALF: none, its slow path re-walks out of registers) + FIN: the 6-KiB tile summary
    // + counters (hrx_kernel.hpp pm_pair_bytes: the planner budgets the same bytes)
    const uint32_t pair_bytes = (uint32_t)pm_pair_bytes(nring, HALF, FIN);
    const uint32_t tab_bytes = GTAB ? 0u : BYTE ? a.byte_bytes : HALF ? a.half_bytes : a.table_bytes;
    const uint32_t ring_base = tab_bytes + pair * pair_bytes;
    const uint32_t scratch_off = ring_base + nring * kPmTileBytes;
    const uint32_t sum_off = scratch_off + (HALF ? 0u : kPmTileBytes);
    const uint32_t ready_off = sum_off + (FIN ? kPmSummaryBytes : 0u), freed_off = ready_off + 4u;
    // FIN: freed2 = ring slots the finisher is done with; sum_ready / sum_freed = tile summaries written / consumed
    const uint32_t freed2_off = ready_off + 8u, sum_ready_off = ready_off + 12u, sum_freed_off = ready_off + 16u;
    // dynamic groups: gq_ready = groups the loader has published, gq = the queue of their indices (16 entries)
    const uint32_t gq_ready_off = ready_off + 20u, gq_off = ready_off + 32u;
    const bool dyn = a.group_counter != nullptr;
    const uint32_t M = a.M, B = a.B;
    const uint32_t ntiles = (M + 63u) >> 6;
    // CHUNKED launch (hrx_kernel_spec.hip): every string is cut into vs_chunks pieces of vs_tiles tiles and each piece is walked as a
    // group of its own — virtual group vg = chunk * vs_groups + real group, chunk-major, so that the chip still writes one compact
    // slab at a time — from the state, previous substr id and end flag the scout / compose launches found for its first row
    // (a.vs_init).  Everything below works on absolute rows, so n, M, padding, the accept state and the error rows keep their meaning.
    const bool vs = FIN && a.vs_init != nullptr;
    const uint32_t gt = vs ? a.vs_tiles : ntiles;                     // tiles per (virtual) group
    auto vg_real = [&](const uint32_t vg) -> uint32_t { return vs ? vg % a.vs_groups : vg; };
    auto vg_tile0 = [&](const uint32_t vg) -> uint32_t { return vs ? (vg / a.vs_groups) * a.vs_tiles : 0u; };
    uint32_t seq = 0, ready_seen = 0;
#if defined(HRX_STAMPS) || defined(HRX_ABLATION)
    // profiling only (HRX_PACE's high half): rotate which 4-KiB class of every slab an XCD's walkers write
    const uint32_t wg_rot = (blockIdx.x & ~7u) | ((blockIdx.x + (a.pace_even >> 16)) & 7u);
#else
    const uint32_t wg_rot = blockIdx.x;
#endif
    const uint32_t g_first = xcd_slot(wg_rot, gridDim.x, (a.debug & kDbgXcdRemap) != 0) * pairs + pair, g_stride = gridDim.x * pairs;
    // j-th group of this pair (j = 0, 1, ..); >= n_groups: there is none.  Static: a fixed stride.  Dynamic (plan_witness_launch:
    // batches of >= 8 long groups per pair): the first group is static, every further one is whatever the pair's LOADER drew from
    // the launch's counter when it got there (it runs ahead of the other two waves) and published in the LDS queue — pairs on
    // faster XCDs simply draw more often.
    auto group_at = [&](const uint32_t j) -> uint32_t {
        if (!dyn || j == 0u) return g_first + j * g_stride;
        ring_wait(gq_ready_off, j);
        return lds_vol_u32(gq_off + (j & 15u) * 4u);
    };
    // The loaders request their pair's first input tile BEFORE the table is staged: the HBM round trip (~2 us) then runs
    // under the staging instead of after it.
    uint32_t first_len = M;   // ... and the walkers their first group's lengths
    if (is_walker && g_first < a.n_groups) first_len = a.lens[min(vg_real(g_first) * 64u + lane, B - 1u)];
    uint4 first_tile[4];
    if (!is_walker && !is_finisher && g_first < a.n_groups) {
        const bool in_pm0 = (a.layout & 2u) != 0;
        const uint32_t gr0 = vg_real(g_first), tr0 = vg_tile0(g_first) * 64u;
        const uint32_t bl = min(gr0 * 64u + lane, B - 1u);
        const uint32_t blk0 = (gr0 * 64u / kPmBlock) * kPmBlock, nb0 = min(kPmBlock, B - blk0);   // this group's block (hrx_lane.h)
        const uint8_t *cptr = in_pm0 ? a.chars + (size_t)blk0 * a.stride + (size_t)(bl - blk0) * 16u : a.chars + (size_t)bl * a.stride;
        const uint32_t row_cap0 = (uint32_t)a.stride - 16u;
        const size_t cmul0 = (a.debug & kDbgInputFromL2) ? (size_t)0 : in_pm0 ? (size_t)nb0 : (size_t)1;
#pragma unroll
        for (uint32_t i = 0; i < 4u; ++i) first_tile[i] = *reinterpret_cast<const uint4 *>(cptr + (size_t)min(tr0 + 16u * i, row_cap0) * cmul0);
    }
    {
        const uint4 *src = WIDE ? reinterpret_cast<const uint4 *>(a.wide_image)
                                : BYTE ? reinterpret_cast<const uint4 *>(a.byte_image)
                                : HALF ? reinterpret_cast<const uint4 *>(a.half_image) : reinterpret_cast<const uint4 *>(a.table_image);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        // (Batching these loads — several per thread in flight before their LDS writes — was tried twice and is SLOWER: the walkers' first
        // row moves from 3.7 to 4.8 us after the launch, 81.5 -> 82.1 us per step; profiles/r03_probes/ab_staging.txt.)
        if (!GTAB)
            for (uint32_t i = threadIdx.x; i < tab_bytes / 16u; i += blockDim.x) dst[i] = src[i];
        if (is_walker && lane == 0) {
            lds_store_u32(ready_off, 0); lds_store_u32(freed_off, 0);
            lds_store_u32(gq_ready_off, 0);
            if (FIN) { lds_store_u32(freed2_off, 0); lds_store_u32(sum_ready_off, 0); lds_store_u32(sum_freed_off, 0); }
        }
    }
    __syncthreads();

    if (is_finisher) {
        // ================================ finisher (FIN) ================================
        const size_t q8 = (M + 7u) / 8u;
        const bool nt_msk = !(a.nt_mix & kNtMixMaskedWb) && !(a.debug & (kDbgNoNtStores | kDbgNoNtMasked));
        MaskCarry mc = {0, 0, 0, 0};
        uint32_t b0_f = 0, blk0_f = 0, nb_f = 0;
        bool active_f = false;
        unsigned char *mp_f = nullptr;
        size_t mstep_f = 0;
        uint32_t f = 0;
        uint32_t byte_sid_prev = 0;   // BYTE: the substr id of the row before the tile (hrx_walk_pm.h: the walker hands tag bytes over, not bitvectors)
        uint32_t mg_sum_prev = 0, mg_ov_row = 0xffffffffu;   // merge pass (WitnessArgs::merge_G): the summed substr id of the row before the tile; lowest cross-group overlap row
        // HOLD (BYTE table — the random-DFA shape of cfg 5, where the optimistic end mask of hrx_lane.h is wrong for ~10 % of all masked
        // rows): the masked rows of the last kHoldF tiles stay in the finisher's registers, so that a fix-up that arrives within
        // kHoldF tiles zeroes them THERE; only what is older was stored already and is repaired at the memory (2-byte scattered stores).
        // Two tiles: a third takes the kernel past the 168 VGPRs three waves per SIMD leave each (8 spills, 0.36 -> 0.41 ms).
        #ifndef HRX_HOLDF
#define HRX_HOLDF 2
#endif
        constexpr int kHoldF = BYTE ? HRX_HOLDF : 0;
        uint32_t hcw[kHoldF ? kHoldF : 1][16], hsid[kHoldF ? kHoldF : 1][16];
        uint64_t hmask[kHoldF ? kHoldF + 1 : 1];
        // A def with ONE substring (WitnessArgs::byte_one_id): a row's id byte is that id or 0 — 64 bits per tile instead of 16 registers — and the registers of the second
        // tile's id bytes hold a THIRD tile's raw bytes: held tile i = raw bytes hcw[0], hcw[1], hsid[0] for i = 0, 1, 2, id bits hidm[i].  (A random DFA's repairs reach back
        // ~330 rows on average: a third of the rows they zero in memory with two held tiles are still in registers with three.)
        uint64_t hidm[kHoldF ? kHoldF + 1 : 1];
        const uint32_t one_id = (BYTE && kHoldF == 2 && !a.summary && a.merge_G == 0u) ? a.byte_one_id : 0u;   // (a merging last pass sums the other groups' ids in: not one id any more)
        uint32_t n_held = 0;
#ifdef HRX_STAMPS
        unsigned long long fk_wait = 0, fk_work = 0, fk_bits = 0, fk_masks = 0, fk_fix = 0, fk_rows = 0;
#endif
        for (uint32_t j = 0;; ++j) {
          const uint32_t vgf = group_at(j);
          if (vgf >= a.n_groups) break;
          const uint32_t gf = vg_real(vgf), tf0 = vg_tile0(vgf);
          uint32_t vs_fwd = 0, vs_dec = 0;   // chunked launch: has the chunk a forward event; what its first deciding tile says about the rows before it
          for (uint32_t tf = tf0; tf < tf0 + gt; ++tf, ++f) {
            const uint32_t t0 = tf << 6;
            if (tf == tf0) {   // a new group: this lane's string
                b0_f = gf * 64u;
                const uint32_t b = b0_f + lane;
                active_f = b < B;
                const uint32_t bc = active_f ? b : B - 1u;     // lanes beyond the batch shadow the last string (same values, same addresses)
                blk0_f = (b0_f / kPmBlock) * kPmBlock; nb_f = min(kPmBlock, B - blk0_f);
                mp_f = reinterpret_cast<unsigned char *>(a.masked) + ((size_t)blk0_f * q8 + (bc - blk0_f)) * 16u;
                mstep_f = (a.debug & kDbgFixedLines) ? (size_t)0 : (size_t)nb_f * 16u;
                mc = MaskCarry{0, 0, 0, 0};
                n_held = 0;
                mg_sum_prev = 0; mg_ov_row = 0xffffffffu;
            }
#ifdef HRX_STAMPS
            const unsigned long long fk_a = clock64();
#endif
            ring_wait(sum_ready_off, f + 1u);
#ifdef HRX_STAMPS
            const unsigned long long fk_b = clock64();
            fk_wait += fk_b - fk_a;
#endif
            uint4 s0 = make_uint4(0, 0, 0, 0);
            if constexpr (!BYTE) s0 = lds_u128(sum_off + lane * 16u);
            const uint4 s1 = lds_u128(sum_off + 1024u + lane * 16u);
            uint32_t sidq[16], cw[16];
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) {
                const uint4 v = lds_u128(sum_off + 2048u + i * 1024u + lane * 16u);
                sidq[4 * i] = v.x; sidq[4 * i + 1] = v.y; sidq[4 * i + 2] = v.z; sidq[4 * i + 3] = v.w;
                const uint4 c = lds_u128(ring_base + (f % nring) * kPmTileBytes + i * 1024u + lane * 16u);
                cw[4 * i] = c.x; cw[4 * i + 1] = c.y; cw[4 * i + 2] = c.z; cw[4 * i + 3] = c.w;
            }
            ring_post_lds(sum_freed_off, f + 1u);   // (the LDS executes it behind the reads above)
            lds_store_u32(freed2_off, f + 1u);
            TileBits tb;
            if constexpr (BYTE) {
                if (tf == tf0) byte_sid_prev = s1.x;   // (a chunk: the id of the transition into its first row; else 0)
                tb = byte_tile_bits(sidq, byte_sid_prev);
                s0 = make_uint4((uint32_t)tb.st, (uint32_t)(tb.st >> 32), (uint32_t)tb.en1, (uint32_t)(tb.en1 >> 32));
            } else {
                tb.st = (uint64_t)s0.x | ((uint64_t)s0.y << 32);
                tb.en1 = (uint64_t)s0.z | ((uint64_t)s0.w << 32);
                tb.ch = (uint64_t)s1.x | ((uint64_t)s1.y << 32);
            }
#ifdef HRX_STAMPS
            const unsigned long long fk_c = clock64();
            fk_bits += fk_c - fk_b;
#endif
            if (a.summary) {   // a pass of a multi-pass config: the combine kernel forms the sums over all defs (hrx_kernel_mp.hip)
                if (active_f) {
                    uint4 *sp = reinterpret_cast<uint4 *>(a.summary) + ((size_t)tf * 5u * B + (b0_f + lane));
                    sp[0] = s0;
#pragma unroll
                    for (uint32_t i = 0; i < 4u; ++i) sp[(size_t)(i + 1u) * B] = make_uint4(sidq[4 * i], sidq[4 * i + 1], sidq[4 * i + 2], sidq[4 * i + 3]);
                }
                continue;
            }
            const uint32_t n_f = s1.z;          // the string's length (<= M; the walker clamps bad lengths)
            if (a.merge_G) {
                // the LAST pass of a multi-pass config: the earlier groups' tile summaries join this group's — what needs ALL defs of a row
                // (lib.rs:467-519, 593-764): any is_start / is_end, Sum(substr_id) and where it changes, two defs flagging one row
                const uint32_t bcl = min(b0_f + lane, B - 1u);
                uint64_t ov_st = 0, ov_en = 0;
                for (uint32_t g = 0; g < a.merge_G; ++g) {
                    const uint4 *sp = reinterpret_cast<const uint4 *>(a.merge_summary[g]) + ((size_t)tf * 5u * B + bcl);
                    const uint4 h = sp[0];
                    const uint64_t gst = (uint64_t)h.x | ((uint64_t)h.y << 32), gen = (uint64_t)h.z | ((uint64_t)h.w << 32);
                    ov_st |= tb.st & gst; ov_en |= tb.en1 & gen;
                    tb.st |= gst; tb.en1 |= gen;
#pragma unroll
                    for (uint32_t i = 0; i < 4u; ++i) {
                        const uint4 v = sp[(size_t)(i + 1u) * B];
                        sidq[4 * i] += v.x; sidq[4 * i + 1] += v.y; sidq[4 * i + 2] += v.z; sidq[4 * i + 3] += v.w;   // byte sums <= 255 (finalize_defs)
                    }
                }
                if (mg_ov_row == 0xffffffffu) {
                    if (ov_st) mg_ov_row = t0 + (uint32_t)ctz64(ov_st);
                    if (ov_en) mg_ov_row = min(mg_ov_row, t0 + (uint32_t)ctz64(ov_en) + 1u);
                }
                uint64_t ch = 0;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const uint32_t x = sidq[q], y = (x << 8) | (q ? (sidq[q - 1] >> 24) : mg_sum_prev);
                    ch |= (uint64_t)nonzero_bytes4(x ^ y) << (4 * q);
                }
                mg_sum_prev = sidq[15] >> 24;
                tb.ch = ch;
                if (tf + 1u == tf0 + gt && active_f) a.merge_ov[b0_f + lane] = mg_ov_row;
            }
            if (vs && tf == tf0) mc.en = s1.w;  // a chunk: the is_end flag that lands on its first row (from the row before it)
            // ---------------- reveal masks: lib.rs:598-764 ----------------
            TileMasks tm = tile_masks<64>(tb, mc, t0, tile_is_exact(t0, n_f, M), rows_below(t0, n_f));
            if (vs) {
                vs_fwd |= tm.fwd;
                if (vs_dec == 0u) vs_dec = tm.dec;
                if (tf + 1u == tf0 + gt && active_f)    // the chunk's last tile: what the stitch launch needs to know about it (hrx_kernel_spec.hip)
                    a.vs_info[(size_t)(vgf / a.vs_groups) * B + (b0_f + lane)] = make_uint2(mc.pend | vs_fwd << 1 | mc.sm << 2 | vs_dec << 3, mc.pend_start);
            }
            if (a.debug & kDbgSkipFixups) tm.fix = 0;
            const uint32_t fix_regs = tm.fix;   // held rows: shadow lanes too (they store the same rows to the same addresses as string B - 1)
            if (!active_f) tm.fix = 0;
#ifdef HRX_STAMPS
            const unsigned long long fk_d = clock64();
            fk_masks += fk_d - fk_c;
#endif
            uint32_t fix_end = t0;     // rows [fix_start, fix_end) were stored already and are fixed at the memory
            if constexpr (kHoldF > 0) {
                // held tile i = tile tf - 1 - i, rows [t0 - 64 (i + 1), t0 - 64 i), kept as what its masked rows are MADE of — raw bytes, id bytes and the
                // 64 mask bits: a fix-up clears mask bits (a handful of instructions per held tile; re-masking finished rows took ~900 cycles of nearly
                // every tile, because some lane of the wave almost always has one), and the rows are assembled when the tile leaves
                fix_end = t0 - n_held * 64u;
                if (fix_regs) {
#pragma unroll
                    for (int i = 0; i < kHoldF + 1; ++i) {
                        const uint32_t base = t0 - 64u * (uint32_t)(i + 1);
                        const uint32_t keep = tm.fix_start <= base ? 0u : min(tm.fix_start - base, 64u);     // rows of the tile in front of fix_start
                        if ((uint32_t)i < n_held) hmask[i] &= keep >= 64u ? ~0ull : ((1ull << keep) - 1ull);
                    }
                }
            }
            uint64_t fixm = __ballot(tm.fix != 0 && tm.fix_start < fix_end);
            while (fixm) {   // an earlier optimistic end_mask = 1 turned out wrong: zero those masked rows (rare with real definitions)
                const int j = __ffsll((unsigned long long)fixm) - 1;
                fixm &= fixm - 1;
                const uint32_t fs = (uint32_t)__builtin_amdgcn_readlane((int)tm.fix_start, j);
                const uint32_t bj = b0_f + (uint32_t)j;
                if constexpr (kHoldF > 0) {
                    // the whole wave zeroes ONE string's rows [fs, fix_end): the rows up to the next octet border two bytes at a time (lanes 0 .. 6), then a
                    // whole octet — 16 bytes — per lane: 512 rows per store instruction (fix_end is a multiple of 64; one row per lane took a store
                    // instruction, each touching eight lines, per 64 rows — and a random DFA's repairs reach back hundreds of rows)
                    const uint32_t o0 = (fs + 7u) >> 3, o1 = fix_end >> 3;
                    const uint32_t rh = fs + lane;
                    if (rh < (o0 << 3)) {
                        const uint32_t rr = (a.debug & kDbgFixToDummy) ? (rh & 63u) : rh;
                        a.masked[((size_t)blk0_f * q8 + (size_t)(rr >> 3) * nb_f + (bj - blk0_f)) * 8u + (rr & 7u)] = 0;
                    }
                    for (uint32_t o = o0 + lane; o < o1; o += 64u) {
                        const uint32_t oo = (a.debug & kDbgFixToDummy) ? (o & 7u) : o;
                        *reinterpret_cast<uint4 *>(reinterpret_cast<unsigned char *>(a.masked) + ((size_t)blk0_f * q8 + (size_t)oo * nb_f + (bj - blk0_f)) * 16u) = make_uint4(0, 0, 0, 0);
                    }
                } else {
                    for (uint32_t r = fs + lane; r < fix_end; r += 64u) {
                        const uint32_t rr = (a.debug & kDbgFixToDummy) ? (r & 63u) : r;
                        a.masked[((size_t)blk0_f * q8 + (size_t)(rr >> 3) * nb_f + (bj - blk0_f)) * 8u + (rr & 7u)] = 0;
                    }
                }
            }
#ifdef HRX_STAMPS
            const unsigned long long fk_e = clock64();
            fk_fix += fk_e - fk_d;
#endif
            // ---------------- masked rows: 8 x 16 B per tile and string, [M/8][B][8] (lib.rs:752-761) ----------------
            // Streamed (non-temporal) unless some string of the wave still has an OPEN optimistic span that reaches into the tile: those rows may have to be zeroed
            // once the span is decided, 16 bytes of a 128-byte line at a time — a repair that finds the line in L2 merges there, one that finds it in memory costs the
            // memory a read-modify-write.  cfg 5 (a random DFA's tags: an event per ~330 rows, half of them a repair reaching back ~5 tiles): 0.716 -> 0.668 ms, and
            // nothing lost without any tagged pair (0.629 both ways); the bench line gains 1.5 % (its planted matches keep spans open for a tile or two), headers3 loses
            // 0.9 % (long spans that DO end): same buffers, policies alternating in one process, profiles/r04_probes/ab_policy.txt.
            auto octets_out = [&](const uint32_t tt, const uint32_t (&c)[16], const uint32_t (&sd)[16], const uint64_t mask, const bool all) {
                const uint32_t km = (a.nt_mix >> 12) & 0xfu;    // (tools only, like bit 0x200 = "never per tile": tools/ab_policy.py)
                const bool nt_tile = nt_msk && !(!(a.nt_mix & kNtMixNoOpenSpan) && __any(mc.pend != 0u && mc.pend_start < ((tt + 1u) << 6))) && !(km != 0u && tt % km == km - 1u);
                const uint32_t mlo = (uint32_t)mask, mhi = (uint32_t)(mask >> 32);
                unsigned char *mp = mp_f + (size_t)tt * 8u * mstep_f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t mbyte = ((k < 4 ? mlo : mhi) >> (8 * (k & 3))) & 0xffu;
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (mbyte) v = masked_octet(c[2 * k], c[2 * k + 1], sd[2 * k], sd[2 * k + 1], mbyte);
                    if ((all || (tt << 6) + (uint32_t)k * 8u < M) && !(a.debug & kDbgSkipMasked)) store16(mp + (size_t)k * mstep_f, v, nt_tile);   // only the octets that exist
                }
            };
            if constexpr (kHoldF > 0) {
              if (one_id) {
                // ---- three held tiles (one substring id): id bytes are rebuilt from the tile's 64 id bits when it leaves
                auto ids_of = [&](const uint64_t idm, uint32_t (&sd)[16]) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) sd[q] = (__umul24((uint32_t)(idm >> (4 * q)) & 0xfu, 0x204081u) & 0x01010101u) * one_id;   // bit i of the nibble -> byte i
                };
                uint32_t sd[16];
                if (n_held == 3u) { ids_of(hidm[2], sd); octets_out(tf - 3u, hsid[0], sd, hmask[2], true); }
#pragma unroll
                for (int q = 0; q < 16; ++q) { hsid[0][q] = hcw[1][q]; hcw[1][q] = hcw[0][q]; hcw[0][q] = cw[q]; }
                hmask[2] = hmask[1]; hmask[1] = hmask[0]; hmask[0] = tm.mask;
                hidm[2] = hidm[1]; hidm[1] = hidm[0];
                uint64_t idm = 0;
#pragma unroll
                for (int q = 0; q < 16; ++q) idm |= (uint64_t)nonzero_bytes4(sidq[q]) << (4 * q);
                hidm[0] = idm;
                if (n_held < 3u) ++n_held;
                if (tf + 1u == tf0 + gt) {   // the group's last tile: the held tiles leave, oldest first
                    if (n_held > 2u) { ids_of(hidm[2], sd); octets_out(tf0 + gt - 3u, hsid[0], sd, hmask[2], false); }
                    if (n_held > 1u) { ids_of(hidm[1], sd); octets_out(tf0 + gt - 2u, hcw[1], sd, hmask[1], false); }
                    ids_of(hidm[0], sd); octets_out(tf0 + gt - 1u, hcw[0], sd, hmask[0], false);
                }
              } else {
                // the oldest held tile (tf - kHoldF) leaves now (a held tile that leaves here is never the group's last one: all 8 octets exist); the others move up
                if (n_held == (uint32_t)kHoldF) octets_out(tf - (uint32_t)kHoldF, hcw[kHoldF - 1], hsid[kHoldF - 1], hmask[kHoldF - 1], true);
#pragma unroll
                for (int i = kHoldF - 1; i > 0; --i) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) { hcw[i][q] = hcw[i - 1][q]; hsid[i][q] = hsid[i - 1][q]; }
                    hmask[i] = hmask[i - 1];
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) { hcw[0][q] = cw[q]; hsid[0][q] = sidq[q]; }
                hmask[0] = tm.mask;
                if (n_held < (uint32_t)kHoldF) ++n_held;
                if (tf + 1u == tf0 + gt) {   // the group's last tile: the held tiles leave, oldest first
#pragma unroll
                    for (int i = kHoldF - 1; i >= 0; --i)
                        if ((uint32_t)i < n_held) octets_out(tf0 + gt - 1u - (uint32_t)i, hcw[i], hsid[i], hmask[i], false);
                }
              }
            } else {
                octets_out(tf, cw, sidq, tm.mask, false);
            }
#ifdef HRX_STAMPS
            { const unsigned long long fk_z = clock64(); fk_work += fk_z - fk_b; fk_rows += fk_z - fk_e; }
#endif
          }
        }
#ifdef HRX_STAMPS
        if (a.stamps && lane == 0) {
            unsigned long long *o = a.stamps + (size_t)(blockIdx.x * pairs + pair) * 16u + 8u;
            o[0] += fk_wait; o[1] += fk_work; o[2] += fk_bits; o[3] += fk_masks; o[4] += fk_fix; o[5] += fk_rows;
        }
#endif
        return;
    }
    if (!is_walker) {
        // ================================ loader ================================
        // string-major input: string b at chars + b*stride; position-major input: 16-byte chunk i of string b at
        // chars + (i*B + b)*16, so one load instruction reads 1 KiB contiguous (coalesced, compact read window).
        //
        // The loader runs RT tiles (RT*4 KiB of its pair's input, 16 B per lane per load) ahead of the walker, in its own
        // registers (192 VGPRs at D = 1 that the kernel owns anyway), over the flattened (group, tile) sequence of the
        // pair: at M <= 1024 practically the whole input of a group is requested in one burst at the start, and the next
        // group's bytes are on their way long before the walker gets there.  It issues nothing but these loads, so the
        // counted s_waitcnt vmcnt(4*(RT-1)) for the oldest tile is exact.  Bytes at or beyond a string's length are
        // read (inside the string's own stride) but never trusted.
        constexpr uint32_t RT = (D == 1 && !FIN) ? 12u : 8u;   // FIN: three waves per SIMD, 168 VGPRs each
        const bool in_pm = (a.layout & 2u) != 0;
        const uint32_t row_cap = (uint32_t)a.stride - 16u;  // last 16-byte chunk that exists for every string
        // The pair's sequence of (group, tile) pairs, discovered group by group: `total` = tiles of the groups known so far,
        // `issue_g` = the group whose tiles are being requested.  Static assignment knows the next group by formula; dynamic
        // assignment draws it from the launch's counter (lane 0, value broadcast) — the compiler waits for every load in flight
        // before it reads the result, which the RT tiles already in registers cover — and publishes it, or the end mark,
        // to the walker and the finisher through the LDS queue.
        uint32_t total = g_first < a.n_groups ? gt : 0u, known = 1u, issue_g = g_first;
        bool ended = total == 0u;
        auto next_group = [&]() {
            uint32_t g;
            if (!dyn) {
                g = g_first + known * g_stride;
            } else {
                uint32_t v = 0;
                if (lane == 0u) v = atomicAdd(a.group_counter, 1u);
                v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
                g = a.group_first_dyn + (v - a.group_base);
                if (g >= a.n_groups) g = 0xffffffffu;
                lds_store_u32(gq_off + (known & 15u) * 4u, g);
                ring_post(gq_ready_off, known);
            }
            if (g < a.n_groups) { issue_g = g; total += gt; ++known; }
            else ended = true;
        };
        uint4 buf[RT * 4u];
        auto issue = [&](const uint32_t q, const uint32_t k) {  // tile q of the pair's sequence (a tile of group issue_g) -> register tile k
            const uint32_t g = vg_real(issue_g), t = vg_tile0(issue_g) + q % gt;
            const uint32_t bl = min(g * 64u + lane, B - 1u);
            const uint32_t blk0 = (g * 64u / kPmBlock) * kPmBlock, nb = min(kPmBlock, B - blk0);   // the group's block (hrx_lane.h)
            const uint8_t *cptr = in_pm ? a.chars + (size_t)blk0 * a.stride + (size_t)(bl - blk0) * 16u : a.chars + (size_t)bl * a.stride;
            // byte offset of chunk-start row r: r * cmul  (kDbgInputFromL2, profiling only: every tile re-reads the hot first lines)
            const size_t cmul_eff = (a.debug & kDbgInputFromL2) ? (size_t)0 : in_pm ? (size_t)nb : (size_t)1;
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) {
                const size_t off = (size_t)min(t * 64u + 16u * i, row_cap) * cmul_eff;
                buf[k * 4u + i] = *reinterpret_cast<const uint4 *>(cptr + off);   // (non-temporal LOADS were tried in round 2: 76.5 vs 75.4 us, dropped)
            }
        };
        // request tile q if it exists; tiles are requested in order, so q == total exactly when q opens a new group
        auto try_issue = [&](const uint32_t q, const uint32_t k) {
            if (q == total && !ended) next_group();
            if (q < total) issue(q, k);
        };
        // The pair's FIRST tile travels alone: requested together with the rest, it queues behind the whole chip's opening
        // burst (~48 MiB) and reaches the walker ~10 us into the launch (in-kernel stamps, tools/kbench) instead of ~1.5.
        if (total > 0) {   // (requested before the table staging; see the kernel's prologue)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) {
                uint4 v = first_tile[i];
                asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
                *(__attribute__((address_space(3))) v4u32 *)(uintptr_t)(ring_base + i * 1024u + lane * 16u) = v4u32{v.x, v.y, v.z, v.w};
            }
            ring_post_lds(ready_off, 1u);
        }
#pragma unroll
        for (uint32_t k = 1; k < RT; ++k) try_issue(k, k);
        try_issue(RT, 0);
        for (uint32_t s0 = 0; s0 < total; s0 += RT) {
#pragma unroll
            for (uint32_t k = 0; k < RT; ++k) {
                const uint32_t sq = s0 + k;
                if (sq < total && sq != 0u) {
                    if (sq >= nring) {
                        ring_wait(freed_off, sq - nring + 1u);                // the walker has read this slot
                        if (FIN) ring_wait(freed2_off, sq - nring + 1u);      // ... and the finisher its raw bytes
                    }
                    const uint32_t slot = ring_base + (sq % nring) * kPmTileBytes;
                    // tile sq was requested RT tiles ago; RT-1 younger tiles (4 loads each) may still be in flight
                    if (sq + RT <= total) {
                        if (RT == 12u) asm volatile("s_waitcnt vmcnt(44)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tail of the sequence: nothing younger is being issued
                    }
#pragma unroll
                    for (uint32_t i = 0; i < 4u; ++i) {
                        uint4 v = buf[k * 4u + i];
                        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));  // after the counted wait, not before
                        *(__attribute__((address_space(3))) v4u32 *)(uintptr_t)(slot + i * 1024u + lane * 16u) = v4u32{v.x, v.y, v.z, v.w};
                    }
                    ring_post_lds(ready_off, sq + 1u);
                    try_issue(sq + RT, k);
                }
            }
        }
        return;
    }

    for (uint32_t j = 0;; ++j) {
        const uint32_t vg = group_at(j);
        if (vg >= a.n_groups) break;
        const uint32_t g = vg_real(vg), tile0 = vg_tile0(vg), chunk = vs ? vg / a.vs_groups : 0u;
        const uint32_t b0 = g * 64u;
        const uint32_t b = b0 + lane;
        const bool active = b < B;
        // Lanes beyond the batch (last group only) are EXACT shadows of string B - 1 — same bytes (the loader clamps the same
        // way), same length, same output addresses — so the record / masked stores need no per-lane predicate: every lane of
        // the wave stores, the shadows re-write the last string's values.  Only the status word and the fix-ups are `active`-only.
        const uint32_t n_raw = j == 0u ? first_len : a.lens[min(b, B - 1u)];
        const bool badlen = n_raw > M;
        const uint32_t n = badlen ? M : n_raw;

        {
            // ================================ walker ================================
            const uint32_t min_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_min_u32(n));
            LaneRegs<D> L;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                L.e[d] = BYTE ? a.dc[d].first_state : HALF ? a.dc[d].half_row_base + a.dc[d].first_state : a.dc[d].first_entry;  // states[d][0] = first_state_val: lib.rs:807
                L.mx[d] = 0;
            }
            L.sid_prev = 0;
            L.ov_row = 0xffffffffu;
            uint32_t vs_en = 0;   // chunked launch: the is_end flag landing on the chunk's first row
            if (vs) {
                // a chunk starts from what the scout / compose launches found for its first row: per def the state, and the substr id and
                // end flag of the transition INTO that row (hrx_kernel_spec.hip)
                const uint32_t *ip = a.vs_init + ((size_t)chunk * B + min(b, B - 1u)) * D;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const uint32_t w = ip[d];
                    const uint32_t row = a.dc[d].row_base + (w & 0xffffu);
                    L.e[d] = BYTE ? (w & 0xffffu) : WIDE ? row << kWideRowShift : row << kNextShift;
                    L.sid_prev += (w >> 16) & 0xffu;
                    vs_en |= (w >> 24) & 1u;
                }
            }
            const uint32_t byte_sid0 = L.sid_prev;   // BYTE: handed to the finisher, which tracks the ids itself
            MaskCarry mc = {0, 0, 0, 0};
            uint32_t dead = 0, accept = 0;
            uint32_t err_pos[D], err_state[D], err_char[D], acc_state[D];
#pragma unroll
            for (int d = 0; d < D; ++d) {
                err_pos[d] = err_state[d] = err_char[d] = 0;
                acc_state[d] = a.dc[d].first_state;  // n == 0
            }
            const uint32_t bc = active ? b : B - 1u;  // idle lanes shadow the last string
            // the group's block of the position-major buffers (hrx_lane.h kPmBlock): nb strings starting at string blk0
            const uint32_t blk0 = (b0 / kPmBlock) * kPmBlock, nb = min(kPmBlock, B - blk0);
            const size_t q4 = (M + 3u) / 4u, q8 = (M + 7u) / 8u;
            // SM (string-major outputs from this kernel: D = 3, which the walker/storer kernel's 128-byte string-tiles do not
            // cover): records [B][pitch][D], masked [B][pitch] — same walk, the lane's own strides
            // a pass of a multi-pass config writes its defs' planes of the caller's [M/4][rec_D][nb][4] buffer (hrx_kernel.hpp)
            const uint32_t RD = a.rec_D ? a.rec_D : (uint32_t)D;
            // record planes in buffers of their own (WitnessArgs::rec_planes): every def's plane is the D = 1 layout [M/4][nb][4] of its block
            const bool planes = !SM && a.rec_planes[0] != nullptr;
            const bool stripes = planes && D == 1 && a.rec_stripes == 2u;      // one def in two row stripes: quad q in buffer q % 2 at slot q / 2
            const size_t slots = stripes ? (q4 + 1u) / 2u : q4;
            unsigned char *rp = SM ? reinterpret_cast<unsigned char *>(a.records) + (size_t)bc * a.rec_pitch * D * 4u
                                : planes ? a.rec_planes[0] + ((size_t)blk0 * slots + (bc - blk0)) * 16u
                                   : reinterpret_cast<unsigned char *>(a.records) + (((size_t)blk0 * q4 * RD + (size_t)a.rec_d0 * nb) + (bc - blk0)) * 16u;
            size_t poff[D];
#pragma unroll
            for (int d = 0; d < D; ++d) poff[d] = planes ? (size_t)(a.rec_planes[d] - a.rec_planes[0]) : (size_t)d * nb * 16u;
            // (kDbgFixedLines, profiling only: every quad / octet of a string lands on the first one — same store instructions, no new lines or pages)
            const size_t rstep = (a.debug & kDbgFixedLines) ? (size_t)0 : SM ? (size_t)16u * D : planes ? (size_t)nb * 16u : (size_t)nb * 16u * RD;  // one quad of rows further: [M/4][D][nb][4]
            unsigned char *mp = SM ? reinterpret_cast<unsigned char *>(a.masked) + (size_t)bc * a.msk_pitch * 2u
                                   : reinterpret_cast<unsigned char *>(a.masked) + ((size_t)blk0 * q8 + (bc - blk0)) * 16u;
            const size_t mstep = (a.debug & kDbgFixedLines) ? (size_t)0 : SM ? (size_t)16u : (size_t)nb * 16u;      // 8 rows further: [M/8][nb][8]
            rp += (size_t)tile0 * 16u * rstep;         // (a chunk: its first quad / octet)
            mp += (size_t)tile0 * 8u * mstep;
            uint4 pend[8];                             // the previous tile's masked rows, not yet stored
            unsigned char *pend_mp = mp;
            bool have_pend = false;
            // HOLD (HALF table, one def — the random-DFA shape of cfg 5, where the optimistic end mask of hrx_lane.h is wrong for
            // ~10 % of all masked rows): the masked rows of the last kHold = 3 tiles (all the 256 VGPRs allow) stay in registers, so that a fix-up that arrives
            // within kHold tiles zeroes them THERE; only what is older was stored already and needs the 2-byte read-modify-writes
            // at the memory that made the fix-ups 37 % of this launch (0.49 vs 0.31 ms with them skipped).
            constexpr int kHold = (HALF && D == 1 && !SM) ? 3 : 0;
            uint4 held[kHold ? kHold : 1][8];
            uint32_t n_held = 0;
            unsigned char *const mp_group = mp;
#pragma unroll
            for (int k = 0; k < 8; ++k) pend[k] = make_uint4(0, 0, 0, 0);

#ifdef HRX_STAMPS   // tools/kbench only: s_memtime ticks this walker spent waiting for input / walking / in the tile-end work
            unsigned long long tk_wait = 0, tk_walk = 0, tk_end = 0;
            const unsigned long long tk_group = clock64(), wall_start = wall_clock64();
#endif
            for (uint32_t t = tile0; t < tile0 + gt; ++t, ++seq) {
                const uint32_t t0 = t << 6;
                const uint32_t slot = ring_base + (seq % nring) * kPmTileBytes;
#if defined(HRX_STAMPS) || defined(HRX_ABLATION)
                if ((a.pace_even & 0xfffu) && !(blockIdx.x & 1u))   // profiling only: hold the walkers of the even workgroups (= even XCDs) back
                    for (uint32_t i = 0; i < (a.pace_even & 0xfffu); ++i) __builtin_amdgcn_s_sleep(1);
#endif
#ifdef HRX_STAMPS
                const unsigned long long tk_a = clock64();
#endif
                ring_wait_seen(ready_off, seq + 1u, ready_seen);
                const uint32_t sf_seen = FIN ? lds_vol_u32(sum_freed_off) : 0u;     // (looked at when the tile is walked: it arrives with the tile's bytes)
                uint4 cq[4];
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i) cq[i] = lds_u128(slot + i * 1024u + lane * 16u);
                ring_post_lds(freed_off, seq + 1u);
#ifdef HRX_STAMPS
                const unsigned long long tk_b = clock64();
#endif

                uint32_t e_start[D];
#pragma unroll
                for (int d = 0; d < D; ++d) e_start[d] = L.e[d];
                uint32_t sidq[16];
                TileBits tb;
                const bool full = (t0 + 64u < min_n);
                const bool do_store = !(a.debug & kDbgSkipRecords);
                // full-line position-major stores stream past L2 (non-temporal); the string-major lane-direct pieces do not (L2 merges them into lines)
                #if defined(HRX_STAMPS) || defined(HRX_ABLATION)
                // profiling only (HRX_PACE bits 15 / 14 / 13 / 12): records stored write-back instead of streaming by the walkers of odd
                // workgroups / of even workgroups / of every workgroup's pairs 0 and 1 / by every walker for every other tile
                const bool wb_probe = ((a.pace_even & 0x8000u) && (blockIdx.x & 1u)) || ((a.pace_even & 0x4000u) && !(blockIdx.x & 1u)) ||
                                      ((a.pace_even & 0x2000u) && pair < 2u) || ((a.pace_even & 0x1000u) && (t & 1u));
#else
                constexpr bool wb_probe = false;
#endif
                const uint32_t wb_k = a.nt_mix & 0xffu;
                const bool wb_tile = wb_k != 0u && (t % wb_k) == wb_k - 1u;   // the streaming / write-back mix of the records (kNtMixDefault)
                const bool nt_rec = !SM && !wb_probe && !wb_tile && !(a.debug & (kDbgNoNtStores | kDbgNoNtRecords)), nt_msk = !SM && !(a.nt_mix & kNtMixMaskedWb) && !(a.debug & (kDbgNoNtStores | kDbgNoNtMasked));
                const bool pend_store = have_pend && !(a.debug & kDbgSkipMasked);
                uint32_t tile_ov = 0, hb = 0;   // WIDE: flag-overlap seen in the tile; bytes >= 128 among the tile's live rows
                // [ceil(M/4)][D][nb][4]: one def's quads of all strings of the block
                GlobalSink<D, SM> sink{rp, poff, rstep, stripes ? (size_t)0 : rstep, stripes ? (size_t)(a.rec_planes[1] - a.rec_planes[0]) : (size_t)0, do_store, nt_rec, nt_msk,
                                       pend, pend_mp, mstep, pend_store, {}};
                if (WIDE) {
                    const uint32_t cwl[16] = {cq[0].x, cq[0].y, cq[0].z, cq[0].w, cq[1].x, cq[1].y, cq[1].z, cq[1].w,
                                              cq[2].x, cq[2].y, cq[2].z, cq[2].w, cq[3].x, cq[3].y, cq[3].z, cq[3].w};
                    if (full) {
                        tb = walk_tile_pm_wide<D, true>(L, cq, a, sink, 0, 0, tile_ov, sidq, acc_state);
#pragma unroll
                        for (int q = 0; q < 16; ++q) hb |= cwl[q];
                        hb &= 0x80808080u;
                    } else {
                        tb = walk_tile_pm_wide<D, false>(L, cq, a, sink, (int)n - (int)t0, (int)M - 1 - (int)t0, tile_ov, sidq, acc_state);
                        const uint32_t live_rows = n > t0 ? min(n - t0, 64u) : 0u;
#pragma unroll
                        for (int q = 0; q < 16; ++q) {   // bytes at or beyond the string's length are not trusted
                            const uint32_t nb = live_rows > 4u * q ? min(live_rows - 4u * q, 4u) : 0u;
                            hb |= cwl[q] & (nb >= 4u ? 0xffffffffu : ((1u << (8u * nb)) - 1u));
                        }
                        hb &= 0x80808080u;
                    }
                } else if constexpr (BYTE) {
                    // (sidq: the tile's TAG bytes here, and no bitvectors — the finisher derives them, byte_tile_bits)
                    tb = TileBits{0, 0, 0};
                    if (full) walk_tile_pm_byte<true>(L, cq, a, sink, 0, 0, sidq, acc_state);
                    else walk_tile_pm_byte<false>(L, cq, a, sink, (int)n - (int)t0, (int)M - 1 - (int)t0, sidq, acc_state);
                } else if (full)
                    tb = walk_tile_pm<D, true, GTAB, HALF>(L, cq, a, sink, 0, 0, t0, sidq, acc_state);
                else
                    tb = walk_tile_pm<D, false, GTAB, HALF>(L, cq, a, sink, (int)n - (int)t0, (int)M - 1 - (int)t0, t0, sidq, acc_state);
                rp = sink.rp;
#ifdef HRX_STAMPS
                const unsigned long long tk_c = clock64();
#endif

                // ---------------- undefined transition (lib.rs:817): rare slow path, re-walk the tile ----------------
                uint32_t newly = 0;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    // WIDE: the dead row absorbs, so the last real chain word tells; a byte >= 128 has no column and was
                    // walked through its masked alias, so such a tile is re-walked as well
                    const bool hit = WIDE ? ((L.mx[d] & kWideRowMask) == a.dc[d].dead_entry || hb != 0)
                                          : BYTE ? L.mx[d] >= a.byte_dead
                                          : HALF ? L.mx[d] >= kHalfDead : L.mx[d] >= a.dc[d].dead_entry;
                    if (!((dead >> d) & 1u) && hit) newly |= 1u << d;
                }
                if (HALF && __any(newly != 0)) {
                    // no scratch area in this variant (a 256-state table leaves 32 KiB of LDS for all the rings): the tile is
                    // re-walked out of the byte registers, fully unrolled
                    const uint32_t cwl[16] = {cq[0].x, cq[0].y, cq[0].z, cq[0].w, cq[1].x, cq[1].y, cq[1].z, cq[1].w,
                                              cq[2].x, cq[2].y, cq[2].z, cq[2].w, cq[3].x, cq[3].y, cq[3].z, cq[3].w};
                    const uint32_t live_rows = n > t0 ? min(n - t0, 64u) : 0u;
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        if ((newly >> d) & 1u) {
                            uint32_t e = e_start[d];
                            bool found = false;
#pragma unroll
                            for (int p = 0; p < 64; ++p) {
                                const uint32_t c = (cwl[p >> 2] >> (8 * (p & 3))) & 0xffu;
                                const uint32_t nx = lds_u16(half_addr(e & 0xffu, c));
                                if (!found && (uint32_t)p < live_rows && nx >= kHalfDead) {
                                    err_pos[d] = t0 + (uint32_t)p;
                                    err_state[d] = (e & 0xffu) - a.dc[d].half_row_base;
                                    err_char[d] = c;
                                    found = true;
                                }
                                e = nx;
                            }
                            dead |= 1u << d;
                        }
                    }
                } else if (__any(newly != 0)) {
                    // the tile's bytes go to this walker's LDS scratch so that the re-walk can index them at run time
#pragma unroll
                    for (uint32_t i = 0; i < 4u; ++i)
                        *(__attribute__((address_space(3))) v4u32 *)(uintptr_t)(scratch_off + lane * 64u + i * 16u) =
                            v4u32{cq[i].x, cq[i].y, cq[i].z, cq[i].w};
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        if ((newly >> d) & 1u) {
                            uint32_t e = e_start[d];
                            const uint32_t live_rows = n > t0 ? min(n - t0, 64u) : 0u;
                            bool found = false;
                            for (uint32_t p = 0; p < live_rows; ++p) {
                                const uint32_t c = smem[scratch_off + lane * 64u + p];
                                uint32_t nx;
                                bool bad;
                                if (WIDE) {
                                    nx = c < 128u ? lds_u32((e & kWideRowMask) | (c << 3)) : a.dc[d].dead_entry;
                                    bad = (nx & kWideRowMask) == a.dc[d].dead_entry;
                                } else if (BYTE) {
                                    nx = lds_u8((e << 8) | c);
                                    bad = nx >= a.byte_dead;
                                } else {
                                    nx = table_at<GTAB>(a, (e & ~kTagMask) | (c << 2));
                                    bad = nx >= a.dc[d].dead_entry;
                                }
                                if (bad) {
                                    err_pos[d] = t0 + p;
                                    err_state[d] = BYTE ? e : (WIDE ? ((e >> kWideRowShift) & 0xffu) : (e >> kNextShift)) - a.dc[d].row_base;
                                    err_char[d] = c;
                                    found = true;
                                    break;
                                }
                                e = nx;
                            }
                            if (found || !WIDE) dead |= 1u << d;
                        }
                    }
                }
                // ---------------- WIDE: two defs flagged the same row somewhere in this tile: find the row (rare) ----------------
                if (WIDE && D > 1 && __any(tile_ov != 0 && L.ov_row == 0xffffffffu)) {
#pragma unroll
                    for (uint32_t i = 0; i < 4u; ++i)
                        *(__attribute__((address_space(3))) v4u32 *)(uintptr_t)(scratch_off + lane * 64u + i * 16u) =
                            v4u32{cq[i].x, cq[i].y, cq[i].z, cq[i].w};
                    if (tile_ov != 0 && L.ov_row == 0xffffffffu) {
                        uint32_t e[D];
#pragma unroll
                        for (int d = 0; d < D; ++d) e[d] = e_start[d];
                        const uint32_t live_rows = n > t0 ? min(n - t0, 64u) : 0u;
                        for (uint32_t p = 0; p < live_rows && L.ov_row == 0xffffffffu; ++p) {
                            const uint32_t c = smem[scratch_off + lane * 64u + p] & 0x7fu;
                            uint32_t T = 0;
#pragma unroll
                            for (int d = 0; d < D; ++d) {
                                e[d] = lds_u32((e[d] & kWideRowMask) | (c << 3));
                                T += e[d];
                            }
                            if (t0 + p + 1u >= M) T &= ~(3u << kWideEndShift);
                            const uint32_t F = T >> kWideStartShift;
                            if (F & 2u) L.ov_row = t0 + p;                       // two is_start flags on row p
                            else if (F & 8u) L.ov_row = t0 + p + 1u;             // two is_end flags on row p+1
                        }
                    }
                }
                // ---------------- accept state: the state at row n (lib.rs:437-457) ----------------
                if (!full && n == t0 + 64u && t + 1 == ntiles) {  // n == M: row n does not exist, s[n] is the live state
#pragma unroll
                    for (int d = 0; d < D; ++d)
                        acc_state[d] = BYTE ? L.e[d] : HALF ? (L.e[d] & 0xffu) - a.dc[d].half_row_base
                                            : (WIDE ? ((L.e[d] >> kWideRowShift) & 0xffu) : (L.e[d] >> kNextShift)) - a.dc[d].row_base;
                }
                if constexpr (FIN) {
                    // ---------------- hand the tile over to the finisher wave: bitvectors, substr-id bytes, the string's length ----------------
                    ring_wait_seen(sum_freed_off, seq, sf_seen);   // it has consumed the previous tile's summary (one summary area per pair)
                    typedef __attribute__((address_space(3))) v4u32 lds_v4u32;
                    if constexpr (BYTE) {   // tag bytes instead of bitvectors; word 0: the substr id of the row before the group's first tile
                        *(lds_v4u32 *)(uintptr_t)(sum_off + 1024u + lane * 16u) = v4u32{byte_sid0, 0u, n, vs_en};
                    } else {
                        *(lds_v4u32 *)(uintptr_t)(sum_off + lane * 16u) = v4u32{(uint32_t)tb.st, (uint32_t)(tb.st >> 32), (uint32_t)tb.en1, (uint32_t)(tb.en1 >> 32)};
                        *(lds_v4u32 *)(uintptr_t)(sum_off + 1024u + lane * 16u) = v4u32{(uint32_t)tb.ch, (uint32_t)(tb.ch >> 32), n, vs_en};
                    }
#pragma unroll
                    for (uint32_t i = 0; i < 4u; ++i)
                        *(lds_v4u32 *)(uintptr_t)(sum_off + 2048u + i * 1024u + lane * 16u) = v4u32{sidq[4 * i], sidq[4 * i + 1], sidq[4 * i + 2], sidq[4 * i + 3]};
                    ring_post_lds(sum_ready_off, seq + 1u);
                    ready_seen = lds_vol_u32(ready_off);      // the next tile's look at the loader's counter
                } else {
                // ---------------- reveal masks: lib.rs:598-764 ----------------
                TileMasks tm = tile_masks<64>(tb, mc, t0, tile_is_exact(t0, n, M), rows_below(t0, n));
                if (a.debug & kDbgSkipFixups) tm.fix = 0;  // profiling only: skip the fix-ups
                const uint32_t fix_regs = tm.fix;   // held rows: shadow lanes too (they store the same rows to the same addresses as string B - 1)
                if (!active) tm.fix = 0;   // (tm.mask stays: a shadow lane stores the same masked rows as string B - 1)
                uint32_t fix_end = t0;     // rows [fix_start, fix_end) were stored already and are fixed at the memory
                if constexpr (kHold > 0) {
                    // held[i] = the masked rows of tile t - 1 - i, rows [t0 - 64 (i + 1), t0 - 64 i): zero what lies at or after fix_start
                    fix_end = t0 - n_held * 64u;
                    if (fix_regs) {
#pragma unroll
                        for (int i = 0; i < kHold; ++i) {
                            if ((uint32_t)i < n_held) {
                                const uint32_t base = t0 - 64u * (uint32_t)(i + 1);
#pragma unroll
                                for (int k = 0; k < 8; ++k) {
                                    const uint32_t row0 = base + 8u * (uint32_t)k;
                                    // u16 index of the first row to zero inside this octet: 0 = all of it, >= 8 = none
                                    const uint32_t keep = tm.fix_start <= row0 ? 0u : min(tm.fix_start - row0, 8u);
                                    uint32_t w[4] = {held[i][k].x, held[i][k].y, held[i][k].z, held[i][k].w};
#pragma unroll
                                    for (uint32_t q = 0; q < 4u; ++q) w[q] = keep > 2u * q + 1u ? w[q] : (keep > 2u * q ? (w[q] & 0xffffu) : 0u);
                                    held[i][k] = make_uint4(w[0], w[1], w[2], w[3]);
                                }
                            }
                        }
                    }
                }
                // An earlier optimistic end_mask = 1 turned out wrong: zero those masked rows (rare with real definitions; a
                // random DFA like cfg 5's takes this path every few tiles, and there each 16-byte piece re-written in a line
                // that has left L2 is a read-modify-write at the memory; a per-lane variant that zeroes whole octets with 16-byte
                // stores was no better).
                uint64_t fixm = __ballot(tm.fix != 0 && tm.fix_start < fix_end);
                while (fixm) {
                    const int j = __ffsll((unsigned long long)fixm) - 1;
                    fixm &= fixm - 1;
                    const uint32_t fs = (uint32_t)__builtin_amdgcn_readlane((int)tm.fix_start, j);
                    const uint32_t bj = b0 + (uint32_t)j;
                    for (uint32_t r = fs + lane; r < fix_end; r += 64u) {
                        const uint32_t rr = (a.debug & kDbgFixToDummy) ? (r & 63u) : r;
                        a.masked[SM ? (size_t)bj * a.msk_pitch + rr : ((size_t)blk0 * q8 + (size_t)(rr >> 3) * nb + (bj - blk0)) * 8u + (rr & 7u)] = 0;
                    }
                }
                // ---------------- masked rows of this tile: 8 x 16 B per string, [M/8][B][8]; stored during the next walk ----------------
                {
                    const uint32_t cw[16] = {cq[0].x, cq[0].y, cq[0].z, cq[0].w, cq[1].x, cq[1].y, cq[1].z, cq[1].w,
                                             cq[2].x, cq[2].y, cq[2].z, cq[2].w, cq[3].x, cq[3].y, cq[3].z, cq[3].w};
                    const uint32_t mlo = (uint32_t)tm.mask, mhi = (uint32_t)(tm.mask >> 32);
                    if constexpr (kHold > 0) {
                        // the oldest held tile (t - kHold) leaves now, as a burst (this kernel's walker has the time: the launch is
                        // memory-bound at 2.4x its chain); the others move up
                        if (n_held == (uint32_t)kHold && !(a.debug & kDbgSkipMasked)) {
                            unsigned char *op = mp_group + (size_t)(t - (uint32_t)kHold) * 8u * mstep;
#pragma unroll
                            for (int k = 0; k < 8; ++k) store16(op + (size_t)k * mstep, held[kHold - 1][k], nt_msk);   // (a held tile is never the last one: all 8 octets exist)
                        }
#pragma unroll
                        for (int i = kHold - 1; i > 0; --i)
#pragma unroll
                            for (int k = 0; k < 8; ++k) held[i][k] = held[i - 1][k];
                        if (n_held < (uint32_t)kHold) ++n_held;
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const uint32_t mbyte = ((k < 4 ? mlo : mhi) >> (8 * (k & 3))) & 0xffu;
                        uint4 v = make_uint4(0, 0, 0, 0);
                        if (mbyte) v = masked_octet(cw[2 * k], cw[2 * k + 1], sidq[2 * k], sidq[2 * k + 1], mbyte);  // lib.rs:752-761
                        if (kHold > 0) held[0][k] = v;
                        else if (D == 1) pend[k] = v;  // leaves during the next tile's walk
                        else if (t0 + (uint32_t)k * 8u < M && !(a.debug & kDbgSkipMasked))
                            store16(mp + (size_t)k * mstep, v, nt_msk);  // D >= 2: the walk needs the registers; store now
                    }
                    pend_mp = mp;
                    mp += 8u * mstep;
                    have_pend = (D == 1) && kHold == 0;
                }
                }
#ifdef HRX_STAMPS
                if (!FIN) asm volatile("" : "+v"(pend[0].x), "+v"(pend[7].w));
                const unsigned long long tk_d = clock64();
                tk_wait += tk_b - tk_a; tk_walk += tk_c - tk_b; tk_end += tk_d - tk_c;
#endif
            }
#ifdef HRX_STAMPS
            if (a.stamps && lane == 0) {
                unsigned long long *o = a.stamps + (size_t)(blockIdx.x * pairs + pair) * 16u;
                o[0] += tk_wait; o[1] += tk_walk; o[2] += tk_end; o[3] += clock64() - tk_group;
                if (j == 0u) { o[4] = wall_entry; o[5] = wall_start; }
                o[6] = wall_clock64();
            }
#endif
            if constexpr (kHold > 0) {   // the held tiles, oldest first; only the octets that exist
                if (!(a.debug & kDbgSkipMasked)) {
#pragma unroll
                    for (int i = kHold - 1; i >= 0; --i) {
                        if ((uint32_t)i < n_held) {
                            const uint32_t tt = ntiles - 1u - (uint32_t)i;
#pragma unroll
                            for (int k = 0; k < 8; ++k)
                                if ((tt << 6) + (uint32_t)k * 8u < M) store16(mp_group + ((size_t)tt * 8u + (size_t)k) * mstep, held[i][k], !(a.debug & (kDbgNoNtStores | kDbgNoNtMasked)));
                        }
                    }
                }
            }
            // the last tile's masked rows (only the octets that exist: [ceil(M/8)][B][8])
            if (have_pend && !(a.debug & kDbgSkipMasked)) {
                const uint32_t t0 = (ntiles - 1u) << 6;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (t0 + (uint32_t)k * 8u < M) store16(pend_mp + (size_t)k * mstep, pend[k], !SM && !(a.debug & (kDbgNoNtStores | kDbgNoNtMasked)));
            }
            // ---------------- per-string status ----------------
            if (active) {
                accept = 0;
#pragma unroll
                for (int d = 0; d < D; ++d) accept |= (acc_state[d] == a.dc[d].accepted_state ? 1u : 0u) << d;
                uint64_t sw;
                if (badlen) {
                    sw = kStatusBadLength;
                } else if (dead) {
                    sw = 0;
#pragma unroll
                    for (int d = D - 1; d >= 0; --d)  // lowest def wins: the reference walks defs in order (lib.rs:806)
                        if ((dead >> d) & 1u) sw = status_invalid((uint32_t)d, err_pos[d], err_state[d], err_char[d]);
                } else if (D > 1 && L.ov_row != 0xffffffffu) {
                    sw = status_overlap(L.ov_row);
                } else {
                    sw = status_ok(accept);
                }
                if (vs) a.vs_status[(size_t)chunk * B + b] = sw;   // a chunk's view of its string; the stitch launch merges them
                else a.status[b] = sw;
            }
        }
    }
}

template <int D, bool GTAB, bool WIDE = false, bool HALF = false, bool SM = false, bool BYTE = false>
static hipError_t launch_pm(const WitnessArgs &a, const LaunchInfo &li, hipStream_t stream) {
    auto k = witness_pm_kernel<D, GTAB, WIDE, HALF, SM, BYTE>;
    static std::atomic<size_t> granted[64];  // per device: the attribute is set on the current device's function
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipError_t e = ensure_lds(k, granted[dev & 63], li.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(li.grid), dim3(64 * li.waves_per_wg), li.lds_bytes, stream, a, (uint32_t)li.nslots);
    return hipGetLastError();
}

hipError_t launch_witness_pm(const WitnessArgs &a, const LaunchInfo &li, hipStream_t stream) {
    if (!(a.layout & 1u)) {   // string-major outputs, D = 3 (plan_witness_launch)
        if (a.D != 3 || li.half || li.gtab) return hipErrorInvalidValue;
        return li.wide ? launch_pm<3, false, true, false, true>(a, li, stream) : launch_pm<3, false, false, false, true>(a, li, stream);
    }
    if (li.byte) return a.D == 1 ? launch_pm<1, false, false, false, false, true>(a, li, stream) : hipErrorInvalidValue;
    if (li.half) return a.D == 1 ? launch_pm<1, false, false, true>(a, li, stream) : a.D == 2 ? launch_pm<2, false, false, true>(a, li, stream) : launch_pm<3, false, false, true>(a, li, stream);
    if (li.wide) return a.D == 1 ? launch_pm<1, false, true>(a, li, stream) : a.D == 2 ? launch_pm<2, false, true>(a, li, stream) : launch_pm<3, false, true>(a, li, stream);
    if (li.gtab) return a.D == 1 ? launch_pm<1, true>(a, li, stream) : a.D == 2 ? launch_pm<2, true>(a, li, stream) : launch_pm<3, true>(a, li, stream);
    return a.D == 1 ? launch_pm<1, false>(a, li, stream) : a.D == 2 ? launch_pm<2, false>(a, li, stream) : launch_pm<3, false>(a, li, stream);
}

}  // namespace hrx
