// hrx_kernel.hip — launch planning and dispatch of the witness kernels (hrx_kernel_sm.hip, hrx_kernel_pm.hip) and the
// small auxiliary kernels: the states-in entry points of lib.rs:825-888 and the field-cell expansion.
#include <hip/hip_runtime.h>

#include "hrx_device.h"
#include "hrx_fr.h"
#include "hrx_kernel.hpp"
#include "hrx_lane.h"

namespace hrx {

// Which of the position-major kernel's stores are write-back instead of streaming.  All-streaming output is not the best this
// memory system does with a launch that writes more than the 256-MB Infinity Cache holds: with about 128 MiB of the records
// stored write-back (every k-th tile's) the bench line runs at 63-67 us in every process instead of 70 or 78 (NOTES_MEASUREMENTS.md §4.1:
// same-process and fresh-process A/Bs, tools/nt_mix_sweep.sh) and D = 3 at 65536 x 1024 B goes 0.72 -> 0.84.  Part of that is the
// cache absorbing lines that the NEXT launch overwrites (the bench re-writes its buffers every step); with outputs rotating over
// 4-8 buffer sets the policy is neutral (81-83 us either way, tools/rotating_outputs.py), and whoever consumes the rows next
// finds that share in the cache.  Not for launches whose whole footprint fits the cache (there streaming wins: 36.1 vs 38.0 us
// at M = 512), nor for the HALF kernel (cfg 5: 0.42 vs 0.43-0.45 ms), nor where k would exceed 8.  The string-major walker/storer
// kernel takes the same policy per 64-row block (cfg 2 string-major: 0.65 -> 0.71 of the peak).
uint32_t plan_nt_mix(const WitnessArgs &a, const LaunchInfo &li) {
    const bool pm = (a.layout & 1u) && li.split == 2 && !li.half, sm_split = !(a.layout & 1u) && li.split == 1;   // (the walker/storer kernel streams full lines too)
    if (!pm && !sm_split) return 0u;
    // The finisher's open-span rule (masked rows of a tile into which an open optimistic span reaches are written back, so that a repair merges in L2) pays where repairs happen — one-def
    // kernels: the bench line + 1.7 %, cfg 5 + 4-8 % — and costs where spans are long and DO end: headers3 65536 x 2048 0.687 with it against 0.721 without, regex2+3 0.745 against 0.749,
    // regex123 0.751 against 0.753 (tools/ab_policy.py on the release kernels, two leases, profiles/r05_probes/ab_policy.txt).  From two defs on the masked rows are streamed.
    const uint32_t flags = pm && a.D >= 2u ? kNtMixNoOpenSpan : 0u;
    const size_t rows = pm ? (size_t)a.M : (size_t)a.rec_pitch;   // string-major: the rows between consecutive strings
    const size_t rec_bytes = (size_t)a.B * rows * 4u * a.D, msk_bytes = (size_t)a.B * a.M * 2u;
    if (rec_bytes + msk_bytes < ((size_t)256 << 20)) return flags;
    const size_t k = (rec_bytes + ((size_t)128 << 20) - 1) / ((size_t)128 << 20);
    return (k < 2 ? 2u : k <= 8 ? (uint32_t)k : 0u) | flags;
}

static bool plan_witness_launch_groups(WitnessArgs &a, int num_cus, LaunchInfo &out);

// A whole config of 4 .. 8 defs through the def-parallel kernel on the CLASS-WIDE tables (hrx_kernel_pmd.hip CW): position-major outputs, a.cw_image / a.table_bytes set.  One group per
// workgroup: D walkers + a combiner wave + a loader.  Same-lease A/B against the passes over groups of three
// defs, 65536 x 2048 / 1024 rows (tools/dn_bench.py, outputs equal bit for bit): D = 4 0.468 against 0.577 ms / 0.220 against 0.285; D = 5 0.581 against 0.637 / 0.293 against 0.319;
// D = 6 0.644 against 0.763 / 0.317 against 0.371; D = 7 0.797 against 0.922 / 0.415 against 0.458.  (With the last def's walker combining — the first version — four and five defs ran no
// faster than their two passes: the combiner, walk + D - 1 merges + reveal mask + masked rows per tile, set the pace of every group.)
bool plan_pmd_cw(WitnessArgs &a, int num_cus, LaunchInfo &out) {
    out = LaunchInfo{};
    a.gs = 64;
    a.n_groups = (uint32_t)(((size_t)a.B + 63) / 64);
    if (!(a.layout & 1u) || !a.cw_image || a.D < 4u || a.D > 8u || (a.debug & kDbgNoDefParallel)) return false;
    const int fin = 1;      // a combiner wave of its own: D + 2 waves (seven and eight defs: nine and ten waves at 168 VGPRs, the combiner's merge loop rolled)
    for (int ns = 4; ns >= 2; --ns) {
        const size_t lds = a.table_bytes + pmd_group_bytes((int)a.D + fin, ns);     // the publishing walkers' areas
        if (lds > kLdsLimit) continue;
        out.split = 5;
        out.wide = 1;
        out.waves_per_wg = (int)a.D + 1 + fin;
        out.nslots = ns;
        out.lds_bytes = lds;
        out.grid = (int)((size_t)a.n_groups < (size_t)num_cus ? (size_t)a.n_groups : (size_t)num_cus);
        if (out.grid < 1) out.grid = 1;
        return true;
    }
    return false;
}

// ... with STRING-MAJOR outputs straight out of the launch (four and five defs, rows in multiples of 16): + a storer wave per group, two LDS sub-tile buffers of [64 strings][D][16 rows]
// and the masked rows' 8-KiB transpose buffer (hrx_kernel_pmd.hip SMO)
bool plan_pmd_cw_sm(WitnessArgs &a, int num_cus, LaunchInfo &out) {
    out = LaunchInfo{};
    a.gs = 64;
    a.n_groups = (uint32_t)(((size_t)a.B + 63) / 64);
    if ((a.layout & 1u) || !a.cw_image || a.D < 4u || a.D > 5u || (a.M % 16u) != 0u || (a.debug & kDbgNoDefParallel)) return false;
    if ((size_t)a.rec_pitch * a.D * 4u * 64u > 0xffffffffull) return false;      // the storer's 32-bit lane offsets span 64 strings' records
    for (int nbuf = 3; nbuf >= 2; --nbuf)      // three sub-tile buffers where LDS has room (four defs): 0.691 against 0.712 ms per 65536 x 2048, the same at x 1024 (profiles/r05_probes/sm_out_of_the_launch.txt)
        for (int ns = 4; ns >= (nbuf == 3 ? 3 : 2); --ns) {
            const size_t lds = a.table_bytes + pmd_group_bytes((int)a.D + 1, ns) + (size_t)nbuf * 64 * (16 * (size_t)a.D + 4) * 4 + 8192;
            if (lds > kLdsLimit) continue;
            a.sm_bufs = (uint32_t)nbuf;
            out.split = 5;
            out.wide = 1;
            out.waves_per_wg = (int)a.D + 3;
            out.nslots = ns;
            out.lds_bytes = lds;
            out.grid = (int)((size_t)a.n_groups < (size_t)num_cus ? (size_t)a.n_groups : (size_t)num_cus);
            if (out.grid < 1) out.grid = 1;
            return true;
        }
    return false;
}

bool plan_witness_launch(WitnessArgs &a, int num_cus, LaunchInfo &out) {
    a.gs = 64;
    a.sm_no_touch = 0;
    a.n_groups = (uint32_t)(((size_t)a.B + 63) / 64);
    return plan_witness_launch_groups(a, num_cus, out);
}

// (a.n_groups given: the chunked launch plans the walk over n_groups x chunks virtual groups)
static bool plan_witness_launch_groups(WitnessArgs &a, int num_cus, LaunchInfo &out) {
    out.gtab = 0;
    out.wide = 0;
    out.half = 0;
    out.byte = 0;
    out.dyn = 0;
    out.spec_tiles = 0;
    out.spec_chunks = 0;
    // DFAs whose fused table leaves no room for the per-wave LDS areas are walked out of global memory (L2-resident)
    const size_t min_stage = (a.layout & 1u) ? pm_pair_bytes(2, false, true) : wave_stage_bytes((int)a.D, 16);
    if (a.table_bytes + min_stage > kLdsLimit || (a.debug & kDbgForceGlobalTable)) out.gtab = 1;
    // position-major loader/walker kernel: from EIGHT groups per walker pair on, the pairs take their groups from a counter
    // instead of a fixed stride — the walkers of odd XCDs run 8-17 % slower than those of even ones (NOTES_MEASUREMENTS.md §4.1), and with a
    // fixed split the launch waits for them.  A group is the unit, so this only pays with many groups per pair: 2^20 x 2048 B
    // (D = 2, 16 groups per pair) 4.68 -> 4.45 ms; with 4 groups per pair the last groups are drawn long before the fast pairs
    // run dry (262144 x 2048 B: 1.188 vs 1.184 ms), and with 16-tile groups the loader's counter access — it waits for its loads
    // in flight — costs more than the balance gains (262144 x 1024 B: 0.320 -> 0.334 ms).  Hence >= 8 groups per pair of >= 32 tiles.
    auto want_dyn = [&](int grid, int pairs) {
        const size_t slots = (size_t)grid * (size_t)pairs;
        if (!(a.layout & 1u) || (a.debug & kDbgNoDynamicGroups)) return 0;
        if (a.debug & kDbgForceDynamicGroups) return (size_t)a.n_groups > slots ? 1 : 0;
        return ((size_t)a.n_groups >= 8 * slots && (a.M + 63u) / 64u >= 32u) ? 1 : 0;
    };
    // "As few walker pairs per workgroup as the batch needs" spreads a small batch over the CUs — but never at the price of a second
    // round: where the table leaves LDS for ONE workgroup per CU, 257-511 groups with one pair per workgroup ran twice as long as with
    // two (pair-step kernel, 24576 x 32768 B: 2.1 vs 1.4 ms).  fixed: table bytes; ring_min(p): the p pairs' smallest rings; wpp: waves per
    // pair; max_waves: per CU.
    auto one_round_pairs = [&](int pairs, const size_t fixed, auto pair_bytes /* (ns) -> one pair's bytes */, const int ns_max, const int ns_min, const int wpp,
                               const int max_waves) {
        auto lds_of = [&](const int p) -> size_t {      // what the loops below pick for p pairs: the deepest ring that fits
            for (int ns = ns_max; ns >= ns_min; --ns)
                if (fixed + (size_t)p * pair_bytes(ns) <= kLdsLimit) return fixed + (size_t)p * pair_bytes(ns);
            return 0;
        };
        for (; pairs < 4; ++pairs) {
            const size_t lds = lds_of(pairs);
            size_t per_cu = lds ? kLdsLimit / lds : 0;
            if (per_cu * (size_t)(wpp * pairs) > (size_t)max_waves) per_cu = (size_t)max_waves / (size_t)(wpp * pairs);
            if (per_cu < 1) per_cu = 1;
            if ((size_t)a.n_groups <= (size_t)num_cus * pairs * per_cu) break;           // one round
            if (!lds_of(pairs + 1)) break;                                                // no room for another pair
        }
        return pairs;
    };
    // ---- CHUNKED launch (hrx_kernel_spec.hip): a batch of at most one group per CU runs for as long as one string's dependent chain
    // (n x 29-50 ns) with three quarters of the walker slots empty.  Cut every string into chunks of 16 tiles, find the chunks' start
    // states (scout + compose) and walk the chunks as groups of their own: the chip is full again.  From two groups per CU on the
    // sequential kernels are bound by the memory system anyway (32768 x 32768 B: 1.3 ms of traffic against 0.95 ms of chain).
    if ((a.layout & 1u) && !out.gtab && !a.summary && !a.merge_G && !a.vs_init && !(a.debug & (kDbgNoSpec | kDbgForceHalf | kDbgForceByte | kDbgForceGlobalTable | kDbgForcePair | kDbgForceDefParallel))) {
        const uint32_t ntiles = (a.M + 63u) / 64u;
        uint32_t tpc = 0;
        if (a.debug & kDbgForceSpec) {
            tpc = 4;
            while (tpc <= kSpecMaxChunkTiles && (ntiles % tpc != 0u || ntiles / tpc > kSpecMaxChunks)) tpc *= 2;
            if (tpc > kSpecMaxChunkTiles || ntiles / tpc < 2u) tpc = 0;
        } else if (ntiles >= 64u && (a.D == 1u ? (size_t)a.n_groups < 2u * (size_t)num_cus
                                               : (size_t)a.n_groups * 4u <= (size_t)num_cus * (a.D == 2u ? 7u : 6u))) {
            // below two groups per CU the sequential kernels run for one or two chains with most walker slots empty (32768-byte strings,
            // 20480 / 28672 strings: D = 1 2.07 / 2.20 ms against 1.05 / 1.41 chunked; D = 2 2.10 / 2.34 against 1.67 / 2.28; D = 3 2.77 / 2.79
            // against 2.37 / 3.23: tools/spec_threshold.sh); from two groups per CU on they are memory-bound and win   // (D = 2: the def-parallel kernel's chain is 1.8 ms per 32768 rows; chunked wins up to 1.25 groups per CU)
            tpc = 16;
            while (tpc <= kSpecMaxChunkTiles && (ntiles % tpc != 0u || ntiles / tpc > kSpecMaxChunks)) tpc *= 2;
            if (tpc > kSpecMaxChunkTiles) tpc = 0;
        }
        for (uint32_t d = 0; d < a.D && d < kMaxDefsPerLaunch; ++d)
            if (a.dc[d].n_rows > 255u) tpc = 0;          // (the scout keeps states as bytes, 0xff = none)
        if (tpc && a.B <= kPmBlock && a.M % 64u == 0u) {
            WitnessArgs v = a;
            v.n_groups = a.n_groups * (ntiles / tpc);
            v.vs_init = reinterpret_cast<const uint32_t *>(&v);   // (any non-null value: the walk over the chunks is planned, not launched)
            v.debug = (a.debug & ~kDbgForceSpec) | kDbgNoPair | kDbgNoDefParallel | kDbgNoSpec;
            LaunchInfo li;
            const uint32_t real_groups = a.n_groups;
            if (plan_witness_launch_groups(v, num_cus, li) && li.split == 2 && !li.half && !li.byte && !li.gtab) {
                out = li;
                out.spec_tiles = (int)tpc;
                out.spec_chunks = (int)(ntiles / tpc);
                a.gs = 64;
                a.n_groups = real_groups;
                return true;
            }
        }
    }
    const uint32_t table_bytes_saved = a.table_bytes;
    struct Restore { WitnessArgs &a; uint32_t v; ~Restore() { a.table_bytes = v; } } restore{a, table_bytes_saved};
    if (out.gtab) a.table_bytes = 0;  // for the LDS budgeting below only; restored on return
    if ((a.layout & 1u) && a.D == 1 && a.byte_image && !(a.debug & (kDbgNoByte | kDbgForceHalf)) &&
        ((out.gtab && !(a.debug & kDbgForceGlobalTable)) || (a.debug & kDbgForceByte))) {
        // ---- loader / walker / finisher kernel on the BYTE table (1-byte next states + the pair tags off the chain): one def of up
        // to 256 states whose 4-byte table does not fit LDS (cfg 5).  Half the HALF table's LDS, which buys what that variant
        // lacks: a finisher wave and a ring of more than one slot (its walker spent 46 % of its cycles in the tile-end work).
        int pairs = 4;
        while (pairs > 1 && (size_t)a.n_groups < (size_t)num_cus * pairs) --pairs;
        pairs = one_round_pairs(pairs, a.byte_bytes, [](int ns) { return pm_pair_bytes((size_t)ns, false, true); }, 4, 2, 3, 12);
        for (; pairs >= 1; --pairs) {
            for (int ns = 4; ns >= 2; --ns) {
                const size_t lds = a.byte_bytes + (size_t)pairs * pm_pair_bytes((size_t)ns, false, true);
                if (lds > kLdsLimit) continue;
                out.split = 2; out.gtab = 0; out.wide = 0; out.half = 0; out.byte = 1;
                out.waves_per_wg = 3 * pairs;
                out.nslots = ns;
                out.lds_bytes = lds;
                const size_t need = ((size_t)a.n_groups + pairs - 1) / pairs;
                size_t per_cu = kLdsLimit / lds;
                if (per_cu * (size_t)(3 * pairs) > 12) per_cu = 12 / (size_t)(3 * pairs);
                if (per_cu < 1) per_cu = 1;
                const size_t cap = (size_t)num_cus * per_cu;
                out.grid = (int)(need < cap ? need : cap);
                if (out.grid < 1) out.grid = 1;
                out.dyn = want_dyn(out.grid, pairs);
                return true;
            }
        }
    }
    if ((a.layout & 1u) && a.half_image && ((out.gtab && !(a.debug & kDbgForceGlobalTable)) || (a.debug & kDbgForceHalf))) {
        // ---- loader/walker kernel on the HALF table (2-byte entries): DFAs of up to 256 states whose 4-byte table does not
        // fit LDS (cfg 5: 256 x 256 -> 128 KiB) stay LDS-resident instead of being walked out of L2 (kDbgForceHalf forces it)
        int pairs = 4;
        while (pairs > 1 && (size_t)a.n_groups < (size_t)num_cus * pairs) --pairs;
        pairs = one_round_pairs(pairs, a.half_bytes, [](int ns) { return pm_pair_bytes((size_t)ns, true, false); }, 4, 1, 2, 8);
        for (; pairs >= 1; --pairs) {
            for (int ns = 4; ns >= 1; --ns) {
                const size_t lds = a.half_bytes + (size_t)pairs * pm_pair_bytes((size_t)ns, true, false);
                if (lds > kLdsLimit) continue;
                out.split = 2; out.gtab = 0; out.wide = 0; out.half = 1;
                out.waves_per_wg = 2 * pairs;
                out.nslots = ns;
                out.lds_bytes = lds;
                const size_t need = ((size_t)a.n_groups + pairs - 1) / pairs;
                size_t per_cu = kLdsLimit / lds;
                if (per_cu * (size_t)(2 * pairs) > 8) per_cu = 8 / (size_t)(2 * pairs);
                if (per_cu < 1) per_cu = 1;
                const size_t cap = (size_t)num_cus * per_cu;
                out.grid = (int)(need < cap ? need : cap);
                if (out.grid < 1) out.grid = 1;
                out.dyn = want_dyn(out.grid, pairs);
                return true;
            }
        }
    }
    // ---- pair-step loader/walker kernel (hrx_kernel_pp.hip): position-major, one def whose PAIR table exists (few byte classes):
    // two bytes per dependent lookup.  Table + per pair a ring of >= 2 slots of 8 KiB (pair indices + raw bytes).
    // Only for batches that leave walker slots empty (< 4 groups per CU): there the launch lasts as long as one string's
    // dependent chain and halving the chain wins (8192 x 32768 B: 1.14 vs 1.57 ms; 16384 x 1023 B: 51.5 vs 58.4 us;
    // 32768: 54.9 vs 59.8 us).  A full chip (>= 4 groups per CU) is bound by the memory system, and the one-byte kernel,
    // with fewer instructions per row, sits on the no-compute mix ceiling: 69.8 vs 73.8 us at 65536 x 1023 B, 131 vs 134 us
    // at 131072 (same-process A/B, profiles/r02_ab_pair_vs_single.txt).
    if ((a.layout & 1u) && a.D == 1 && a.pair_image &&
        ((size_t)a.n_groups < (size_t)num_cus * 4 || (a.debug & kDbgForcePair)) &&
        !(a.debug & (kDbgNoPair | kDbgForceNarrow | kDbgForceWide | kDbgForceHalf | kDbgForceGlobalTable))) {
        // as few pairs per workgroup as cover the batch in ONE round (small batches spread over the CUs); the 76-KiB table leaves room
        // for one workgroup per CU, so "fewer pairs" must not mean "a second round" (24576 x 32768 B ran 2.1 ms with one pair per CU)
        int pairs = 4;
        while (pairs > 1 && (size_t)a.n_groups <= (size_t)num_cus * (pairs - 1)) --pairs;
        for (; pairs >= 1; --pairs) {
            for (int ns = 4; ns >= 2; --ns) {
                const size_t lds = a.pair_bytes + (size_t)pairs * pp_pair_bytes((size_t)ns);
                if (lds > kLdsLimit) continue;
                out.split = 6; out.gtab = 0; out.wide = 0; out.half = 0;
                out.waves_per_wg = 3 * pairs;   // walker + loader + finisher
                out.nslots = ns;
                out.lds_bytes = lds;
                const size_t need = ((size_t)a.n_groups + pairs - 1) / pairs;
                size_t per_cu = kLdsLimit / lds;
                if (per_cu * (size_t)(3 * pairs) > 12) per_cu = 12 / (size_t)(3 * pairs);
                if (per_cu < 1) per_cu = 1;
                const size_t cap = (size_t)num_cus * per_cu;
                out.grid = (int)(need < cap ? need : cap);
                if (out.grid < 1) out.grid = 1;
                return true;
            }
        }
    }
    // ---- def-parallel loader/walker kernel (hrx_kernel_pmd.hip): position-major, D >= 2 on the WIDE table, batches of at
    // most two groups per CU — one walker wave per def, so that a group advances at the single-def rate
    if ((a.layout & 1u) && a.D >= 2 && a.wide_image && !out.gtab &&
        !(a.debug & (kDbgNoDefParallel | kDbgForceNarrow | kDbgForceHalf | kDbgForceGlobalTable)) &&
        a.B <= kPmBlock &&   // one block of the position-major buffers (hrx_lane.h): its strides are the whole batch's
        ((size_t)a.n_groups <= (size_t)num_cus * 2 || (a.debug & kDbgForceDefParallel))) {
        const int G = (size_t)a.n_groups <= (size_t)num_cus && !(a.debug & kDbgForceDefParallel) ? 1 : 2;
        // a combiner wave of its own (FIN): the last def's walker otherwise carries walk + D - 1 merges + reveal mask + masked rows (2.36 of cfg 4's 2.9 ms with every record store
        // compiled out).  kTunePmdFin* (hrx_ctx_set_option HRX_OPT_PMD_COMBINER_WAVE) forces it on / off; the default is what profiles/r06_probes/cfg4_fin_ab.txt measured.
        // Same lease, same input, cfg 4 (tools/planes_ab.py): interleaved records 2.78 ms without / 3.04 with the combiner wave; record planes 2.69 without / 2.53 with — where the records'
        // write stream is what bounds the launch, two more waves per CU only disturb it; where the planes spread it over the classes, the combiner's chain is the bound that is left.
        const bool fin = (a.tune & kTunePmdFinMask) == kTunePmdFinOn || ((a.tune & kTunePmdFinMask) == 0u && a.rec_planes[0] != nullptr);
        for (int ns = 4; ns >= 2; --ns) {
            const size_t lds = a.table_bytes + (size_t)G * (fin ? pmd_fin_group_bytes((int)a.D, ns) : pmd_group_bytes((int)a.D, ns));
            if (lds > kLdsLimit) continue;
            out.split = 5;
            out.wide = 1;
            out.pmd_fin = fin ? 1 : 0;
            out.waves_per_wg = G * ((int)a.D + 1 + (fin ? 1 : 0));
            out.nslots = ns;
            out.lds_bytes = lds;
            const size_t need = ((size_t)a.n_groups + G - 1) / G;
            out.grid = (int)(need < (size_t)num_cus ? need : (size_t)num_cus);
            if (out.grid < 1) out.grid = 1;
            return true;
        }
    }
    // string-major D = 3 (no walker/storer kernel: its string-tiles are 128 bytes = 32 / 16 rows of 1 / 2 defs): the
    // loader/walker kernel with the lane's own string-major strides, 5x the one-wave kernel (NOTES_MEASUREMENTS.md §3.4)
    const bool sm3 = !(a.layout & 1u) && a.D == 3 && a.M % 8u == 0 && !out.gtab && !(a.debug & kDbgForceOneWave);
    if ((a.layout & 1u) || sm3) {
        // ---- loader/walker kernel: table + per pair a ring of up to 4 input tiles (4 KiB each)
        int pairs = 4;
        while (pairs > 1 && (size_t)a.n_groups < (size_t)num_cus * pairs) --pairs;
        {
            const bool fin = (a.layout & 1u) != 0;
            const int wpp0 = fin ? 3 : 2;
            pairs = one_round_pairs(pairs, a.table_bytes, [fin](int ns) { return pm_pair_bytes((size_t)ns, false, fin); }, 4, 2, wpp0, 4 * wpp0);
        }
        for (; pairs >= 1; --pairs) {
            for (int ns = 4; ns >= 2; --ns) {
                const size_t lds = a.table_bytes + (size_t)pairs * pm_pair_bytes((size_t)ns, false, (a.layout & 1u) != 0);   // position-major outputs: the loader finishes the tiles
                if (lds > kLdsLimit) continue;
                out.split = 2;
                // WIDE table: ~4x fewer instructions per row at D = 3; every D >= 2 batch takes it (same-box A/B
                // with spill-free kernels: 1.20 vs 1.33 ms at 262144 x 2048 B, D = 2; 3.48 vs 4.59 ms at 32768 x 32768 B, D = 3).
                // D = 1 gains nothing: its walk is LDS-latency-bound either way.
                out.wide = (a.wide_image && !out.gtab && !(a.debug & kDbgForceNarrow) && (a.D >= 2 || (a.debug & kDbgForceWide))) ? 1 : 0;
                const int wpp = (a.layout & 1u) ? 3 : 2;       // position-major outputs: walker + loader + finisher per pair (hrx_kernel_pm.hip)
                out.waves_per_wg = wpp * pairs;
                out.nslots = ns;
                out.lds_bytes = lds;
                const size_t need = ((size_t)a.n_groups + pairs - 1) / pairs;
                size_t per_cu = kLdsLimit / lds;               // LDS
                if (per_cu * (size_t)(wpp * pairs) > (size_t)(4 * wpp)) per_cu = (size_t)(4 * wpp) / (size_t)(wpp * pairs);  // one pair per SIMD (VGPRs)
                if (per_cu < 1) per_cu = 1;
                const size_t cap = (size_t)num_cus * per_cu;
                out.grid = (int)(need < cap ? need : cap);
                if (out.grid < 1) out.grid = 1;
                out.dyn = want_dyn(out.grid, pairs);   // (0 for the string-major D = 3 use of this kernel: want_dyn checks the layout)
                return true;
            }
        }
        return false;
    }
    // ---- walker/storer kernel on the BYTE table: string-major outputs of one def of up to 256 states whose 4-byte table does not fit
    // LDS (cfg 5; it used to take the position-major kernel + a transpose launch: 0.23 of peak)
    if (!(a.layout & 1u) && a.D == 1 && a.M % 8u == 0 && a.byte_image && out.gtab && !(a.debug & (kDbgNoByte | kDbgForceHalf | kDbgForceGlobalTable | kDbgForceOneWave))) {
        const size_t slot = 64 * 128 + 64 * 8 + 64 * 32, fixed = 16 + 256;
        int pairs = 4;
        while (pairs > 1 && (size_t)a.n_groups < (size_t)num_cus * pairs) --pairs;
        pairs = one_round_pairs(pairs, a.byte16_bytes, [=](int ns) { return (size_t)ns * slot + fixed; }, 4, 2, 2, 8);
        for (; pairs >= 1; --pairs) {
            if (a.byte16_bytes + pairs * (2 * slot + fixed) > kLdsLimit) continue;
            size_t ns = (kLdsLimit - a.byte16_bytes - pairs * fixed) / (pairs * slot);
            if (ns > 4) ns = 4;
            out.split = 1; out.gtab = 0; out.byte = 1;
            a.sm_no_touch = 1;   // (hrx_kernel_sm.hip: the storer's L2 warm-up costs this shape a second pass over the input)
            out.waves_per_wg = 2 * pairs;
            out.nslots = (int)ns;
            out.lds_bytes = a.byte16_bytes + pairs * (ns * slot + fixed);
            const size_t need = ((size_t)a.n_groups + pairs - 1) / pairs;
            out.grid = (int)(need < (size_t)num_cus ? need : (size_t)num_cus);
            if (out.grid < 1) out.grid = 1;
            return true;
        }
    }
    // ---- walker/storer kernel: D in {1,2}, rows in multiples of 8, ring of >= 2 slots per pair
    if ((a.D == 1 || a.D == 2) && a.M % 8u == 0 && !(a.debug & kDbgForceOneWave) && !out.gtab) {
        const size_t slot = 64 * 128 + 64 * 8 + 64 * (a.D == 1 ? 32 : 16), fixed = 16 + 256;  // + the storer's LDS-DMA sink
        int pairs = 4;
        while (pairs > 1 && (size_t)a.n_groups < (size_t)num_cus * pairs) --pairs;  // small batches: spread over the CUs
        pairs = one_round_pairs(pairs, a.table_bytes, [=](int ns) { return (size_t)ns * slot + fixed; }, 4, 2, 2, 8);
        for (; pairs >= 1; --pairs) {
            if (a.table_bytes + pairs * (2 * slot + fixed) > kLdsLimit) continue;
            size_t ns = (kLdsLimit - a.table_bytes - pairs * fixed) / (pairs * slot);
            if (ns > 4) ns = 4;
            out.split = 1;
            out.waves_per_wg = 2 * pairs;
            out.nslots = (int)ns;
            out.lds_bytes = a.table_bytes + pairs * (ns * slot + fixed);
            const size_t need = ((size_t)a.n_groups + pairs - 1) / pairs;
            const size_t wgs_per_cu = kLdsLimit / out.lds_bytes < 1 ? 1 : kLdsLimit / out.lds_bytes;
            const size_t cap = (size_t)num_cus * (wgs_per_cu > 4 ? 4 : wgs_per_cu);
            out.grid = (int)(need < cap ? need : cap);
            if (out.grid < 1) out.grid = 1;
            return true;
        }
    }
    out.split = 0;
    out.nslots = 0;
    // group size: 64 strings per wave; smaller groups only for batches that would otherwise leave CUs without a wave
    // (two half-empty waves per SIMD were measured slower than one full one: the walk is issue-bound, not latency-bound)
    uint32_t gs = 64;
    if (a.debug & kDbgGroups32) gs = 32;
    while (gs > 16 && ((size_t)a.B + gs - 1) / gs < (size_t)num_cus) gs >>= 1;
    a.gs = gs;
    a.n_groups = (uint32_t)(((size_t)a.B + gs - 1) / gs);
    const size_t per_wave = wave_stage_bytes((int)a.D, gs);
    int waves = 0;
    for (int w = 8; w >= 1; --w) {
        if (a.table_bytes + per_wave * w <= kLdsLimit) { waves = w; break; }
    }
    if (!waves) return false;
    // fewer waves per workgroup when the batch cannot feed every CU otherwise
    while (waves > 1 && (size_t)a.n_groups < (size_t)num_cus * waves) --waves;
    out.waves_per_wg = waves;
    out.lds_bytes = a.table_bytes + per_wave * waves;
    const size_t wgs_per_cu = kLdsLimit / out.lds_bytes < 1 ? 1 : kLdsLimit / out.lds_bytes;
    const size_t max_wg = 8 / (size_t)waves < 1 ? 1 : 8 / (size_t)waves;  // <= 8 waves per CU: 2 per SIMD at 157 VGPRs
    const size_t need = ((size_t)a.n_groups + waves - 1) / waves;
    const size_t cap = (size_t)num_cus * (wgs_per_cu > max_wg ? max_wg : wgs_per_cu);
    out.grid = (int)(need < cap ? need : cap);
    if (out.grid < 1) out.grid = 1;
    return true;
}

__global__ void zero_u32_kernel(uint32_t *p) { *p = 0u; }
hipError_t launch_zero_u32(uint32_t *p, hipStream_t stream) {
    hipLaunchKernelGGL(zero_u32_kernel, dim3(1), dim3(1), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_witness(const WitnessArgs &a, const LaunchInfo &li, hipStream_t stream) {
    if (li.split == 6) return launch_witness_pp(a, li, stream);
    if (li.split == 5) return launch_witness_pmd(a, li, stream);
    return li.split == 2 ? launch_witness_pm(a, li, stream) : launch_witness_sm(a, li, stream);
}

// ---------------------------------------------------------------------------------------------
// states-in entry points (lib.rs:825-888): one thread per (def, row) looks the pair (s[i], s[i+1]) up.
// ---------------------------------------------------------------------------------------------
struct PairArgs {     // one def per launch
    const uint64_t *states;   // this def's n + 1 states
    uint64_t n;
    const uint16_t *pt;
    uint32_t ns;
    uint16_t *tags;           // this def's n tags
};

__global__ void pair_tags_kernel(const PairArgs a) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n) return;
    const uint64_t cur = a.states[r], next = a.states[r + 1];
    uint16_t t = 0;
    if (cur < a.ns && next < a.ns) t = a.pt[cur * a.ns + next];
    a.tags[r] = t;
}

hipError_t launch_pair_tags(const uint64_t *states, size_t n, uint32_t D, const uint16_t *const *pair_tags,
                            const uint32_t *n_states, uint16_t *tags, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    for (uint32_t d = 0; d < D; ++d) {
        PairArgs a{states + (size_t)d * (n + 1), n, pair_tags[d], n_states[d], tags + (size_t)d * n};
        hipLaunchKernelGGL(pair_tags_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
    }
    return hipGetLastError();
}

struct EndpointDefArgs {     // one def per launch
    const uint64_t *states, *substr_ids;
    uint64_t n;
    const uint8_t *member;
    uint32_t n_states, n_substrs, id_offset;
    uint8_t *flags;
};

__global__ void endpoint_flags_kernel(const EndpointDefArgs a) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n) return;
    const uint64_t sid = a.substr_ids[r];
    uint8_t f = 0;
    if (sid != 0) {  // lib.rs:861-866, 874-879
        const uint64_t j = sid - a.id_offset;
        const uint64_t cur = a.states[r], next = a.states[r + 1];
        if (j < a.n_substrs) {
            if (cur < a.n_states) f |= a.member[j * a.n_states + cur] & 1;
            if (next < a.n_states) f |= a.member[j * a.n_states + next] & 2;
        }
    }
    a.flags[r] = f;
}

hipError_t launch_endpoint_flags(const EndpointArgs &a, const uint8_t *const *member, const uint32_t *dims, hipStream_t stream) {
    if (a.n == 0) return hipSuccess;
    for (uint32_t d = 0; d < a.D; ++d) {
        EndpointDefArgs e{a.states + (size_t)d * (a.n + 1), a.substr_ids + (size_t)d * a.n, a.n, member[d], dims[3 * d], dims[3 * d + 1], dims[3 * d + 2],
                          a.flags + (size_t)d * a.n};
        hipLaunchKernelGGL(endpoint_flags_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, stream, e);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// compact witness -> field cells.  A pair of threads per (string, row) expands the row's integers into what
// `Value::known(F::from(v))` holds for every advice cell the reference assigns (lib.rs:339-418, 473-519) and for its two
// result columns (lib.rs:752-771), F = bn256::Fr in Montgomery form (hrx_fr.h), column-major [col][string][row][4 limbs].
// Write-bound: 32 B per cell x (4 + 4 D) cells per row.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kFrJ = 4u;                                  // store instructions (32 rows each) per wave and column
constexpr uint32_t kFrRowsPerBlock = 4u * 32u * kFrJ;
__global__ __launch_bounds__(256) void fr_columns_kernel(const FrArgs a) {
    // thread = (row, half of the 32-byte cell): both lanes of a pair compute the cell, each stores its 16 bytes, so that a
    // store instruction writes 1 KiB of full lines (one lane per row stored half of every 32 bytes per instruction)
    // Every value a row holds is small — bytes, states, ids, flags — and F::from(v) costs ~200 VALU operations: the cells of v < 256 come out of an LDS table the block's 256 threads
    // build first (one entry each; 8 KiB), anything larger (states of DFAs beyond 256 states) is computed in place.  Computed per cell by both lanes of a pair the launch was VALU-bound
    // at 0.68 of the HBM peak.
    __shared__ uint4 frtab[512];
    {
        uint32_t w[8];
        fr_from_u32(threadIdx.x, w, a.canonical != 0);
        frtab[2u * threadIdx.x] = make_uint4(w[0], w[1], w[2], w[3]);
        frtab[2u * threadIdx.x + 1u] = make_uint4(w[4], w[5], w[6], w[7]);
    }
    __syncthreads();
    const uint32_t half = threadIdx.x & 1u;
    const uint32_t bi = blockIdx.y;              // index inside the requested range
    const uint32_t b = a.b_begin + bi;
    const uint32_t n = min(a.lens[b], a.M);
    const bool pm = (a.layout & 1u) != 0, in_pm = (a.layout & 2u) != 0;
    // a wave owns 32 kFrJ consecutive rows and writes them column by column: kFrJ store instructions in a row put 4 KiB
    // of ONE column down before the next column's turn (interleaving the columns instruction by instruction left every stream in
    // 1-KiB pieces)
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t rbase = blockIdx.x * kFrRowsPerBlock + wave * (32u * kFrJ) + (lane >> 1);
    const uint32_t blk0 = (b / kPmBlock) * kPmBlock, nb = min(kPmBlock, a.B - blk0), bl = b - blk0;   // the string's block of the position-major buffers
    const size_t col_cells = (size_t)a.col_cells;   // cells per column
    uint64_t *out0 = a.cells + ((size_t)bi * a.M + rbase) * 4u + half * 2u;
    auto put_rows = [&](const uint32_t col, const uint32_t (&v)[kFrJ]) {
#pragma unroll
        for (uint32_t j = 0; j < kFrJ; ++j) {
            if (rbase + j * 32u < a.M) {
                uint4 cell;
                if (v[j] < 256u) {
                    cell = frtab[2u * v[j] + half];
                } else {
                    uint32_t w[8];
                    fr_from_u32(v[j], w, a.canonical != 0);
                    cell = half ? make_uint4(w[4], w[5], w[6], w[7]) : make_uint4(w[0], w[1], w[2], w[3]);
                }
                unsigned char *p = reinterpret_cast<unsigned char *>(out0 + (size_t)col * col_cells * 4u + (size_t)j * 32u * 4u);
                store16_nt(p, cell);    // streaming: nothing reads the cells back here
            }
        }
    };
    uint32_t live[kFrJ], c[kFrJ];
#pragma unroll
    for (uint32_t j = 0; j < kFrJ; ++j) {
        const uint32_t r = min(rbase + j * 32u, a.M - 1u);
        live[j] = (rbase + j * 32u) < n ? 1u : 0u;
        c[j] = 0;
        if (live[j]) c[j] = in_pm ? a.chars[(size_t)blk0 * a.stride + ((size_t)(r >> 4) * nb + bl) * 16u + (r & 15u)] : a.chars[(size_t)b * a.stride + r];
    }
    put_rows(0, live);   // char_enable                        lib.rs:342,346
    put_rows(1, c);      // characters                         lib.rs:343,347
    for (uint32_t d = 0; d < a.D; ++d) {
        uint32_t rec[kFrJ], f[kFrJ];
#pragma unroll
        for (uint32_t j = 0; j < kFrJ; ++j) {
            const uint32_t r = min(rbase + j * 32u, a.M - 1u);
            if (a.rec_planes[0]) {      // (position-major) quad q of def d: buffer (q % R) * D + d, slot q / R of the string's block
                const uint32_t R = a.rec_stripes == 2u ? 2u : 1u, q = r >> 2, q4 = (a.M + 3u) / 4u, slots = (q4 + R - 1u) / R;
                rec[j] = a.rec_planes[(q % R) * a.D + d][(size_t)blk0 * slots * 4u + ((size_t)(q / R) * nb + bl) * 4u + (r & 3u)];
            } else
            rec[j] = pm ? a.records[((size_t)blk0 * ((a.M + 3u) / 4u) * a.D + ((size_t)(r >> 2) * a.D + d) * nb + bl) * 4u + (r & 3u)]
                        : a.records[((size_t)b * a.rec_pitch + r) * a.D + d];
        }
#pragma unroll
        for (uint32_t j = 0; j < kFrJ; ++j) f[j] = rec[j] & 0xffffu;
        put_rows(2 + 4 * d, f);            // states[d]         lib.rs:390,415
#pragma unroll
        for (uint32_t j = 0; j < kFrJ; ++j) f[j] = (rec[j] >> 16) & 0xffu;
        put_rows(3 + 4 * d, f);            // substr_ids[d]     lib.rs:394,405
#pragma unroll
        for (uint32_t j = 0; j < kFrJ; ++j) f[j] = (rec[j] >> 24) & 1u;
        put_rows(4 + 4 * d, f);            // start_enable[d]   lib.rs:483-491
#pragma unroll
        for (uint32_t j = 0; j < kFrJ; ++j) f[j] = (rec[j] >> 25) & 1u;
        put_rows(5 + 4 * d, f);            // end_enable[d]     lib.rs:502-511
    }
    uint32_t mk[kFrJ], f[kFrJ];
#pragma unroll
    for (uint32_t j = 0; j < kFrJ; ++j) {
        const uint32_t r = min(rbase + j * 32u, a.M - 1u);
        mk[j] = pm ? a.masked[((size_t)blk0 * ((a.M + 7u) / 8u) + (size_t)(r >> 3) * nb + bl) * 8u + (r & 7u)] : a.masked[(size_t)b * a.msk_pitch + r];
    }
#pragma unroll
    for (uint32_t j = 0; j < kFrJ; ++j) f[j] = mk[j] & 0xffu;
    put_rows(2 + 4 * a.D, f);              // masked_characters  lib.rs:752-757
#pragma unroll
    for (uint32_t j = 0; j < kFrJ; ++j) f[j] = mk[j] >> 8;
    put_rows(3 + 4 * a.D, f);              // all_substr_ids     lib.rs:758-761
}

hipError_t launch_fr_columns(const FrArgs &a, hipStream_t stream) {
    if (a.b_count == 0 || a.M == 0) return hipSuccess;
    hipLaunchKernelGGL(fr_columns_kernel, dim3((a.M + kFrRowsPerBlock - 1u) / kFrRowsPerBlock, a.b_count), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace hrx
