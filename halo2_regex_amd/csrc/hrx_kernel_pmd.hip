// hrx_kernel_pmd.hip — DEF-PARALLEL position-major kernel: D >= 2 batches that leave walker slots empty (cfg 4 gives a GPU
// 32768 strings = 2 groups per CU, and a group's launch time is one string's serial walk).
//
// With every def walked by the same lane a row costs ~100 ns at D = 3 (three dependent chains share one wave's issue
// slots); a single def walks at ~50 ns per row.  Here a group of 64 strings gets D walker waves — one per def — and one
// loader: each walker runs the D = 1 walk over its own def's rows of the WIDE table and stores its own record plane, so
// the group advances at the single-def rate.  What the defs share is the reveal mask (lib.rs:598-764 works on the SUMS of
// substr ids and flags over the defs): per tile every walker but the last publishes its start / end bitvectors and its
// byte-per-row substr ids in LDS (80 B per lane), and the last def's walker — the combiner — ORs the flags, adds the ids
// (one packed add per four rows), derives the id-changed bits, runs tile_masks and stores the masked rows.  Exact: nothing
// is speculated.  Status pieces (first undefined transition of a def, accept state) are merged by the combiner as well.
//   waves of a workgroup: G groups x D walkers, then G loaders (G = 2: 8 waves at D = 3, 6 at D = 2; 256 VGPRs each).
//   LDS: table | per group: input ring (nring x 4 KiB, freed when all D walkers have taken the tile; slow paths re-walk out
//   of the slot before freeing it) | per walker: 2 summary slots x 5 KiB, a 2-KiB status piece | counters.
#include <hip/hip_runtime.h>

#include "hrx_device.h"
#include "hrx_walk_pm.h"

namespace hrx {

// STRING-MAJOR outputs straight out of the def-parallel launch (SMO; four and five defs): a walker's quads — four rows of its def, 16 B per lane — go into an LDS SUB-TILE
// [string][def][16 rows] (S = 16 D + 4 dwords per string: the lanes' b128 writes meet no bank twice) instead of its plane in memory, and a STORER wave per group moves a finished
// sub-tile out: a string's 16 rows x D records are 64 D contiguous bytes of the caller's [B][pitch][D] buffer, 16 bytes per lane and store, 64 consecutive pieces per instruction (three
// strings' runs at D = 5).  Two or three sub-tile buffers (what LDS has room for): the walkers fill one while the storer drains another.  What the walkers pay is what bounds the launch (their table chain is LDS
// latency, ~6400 cycles of a tile's ~10500): every blocking look at a counter is an LDS round trip on that chain (~90 cycles), every write an issue slot.  Measured per 65536 x 1024 at
// D = 5 (the passes + transposer this replaces, which moved the records three times: 0.93 ms):
//   1. b128 writes, a storer that read four dwords and stored, piece after piece (80 dependent round trips per sub-tile)                                 0.545 ms
//   2. sub-tiles in OUTPUT order [string][16 rows][def] (four 4-byte writes per quad, 4-way bank conflicts), the storer a ds_read_b128 per piece         0.459 ms
//   3. the same with the WALKERS storing a D-th of the sub-tile each, no storer wave (5 + 5 counter waits per sub-tile and walker)                        0.490 ms
//      (ablation build, 3.: no stores 0.458, no sub-tile writes 0.462, neither 0.404 — against 0.28 ms for the position-major launch with everything)
//   4. this: b128 writes; the storer reads ALL of a sub-tile's dwords before its first store (one round trip, offsets loop-invariant), looks at the walkers' counters
//      with one round trip, stores through an SGPR base; the walkers post without waiting for their writes (LDS executes a wave's operations in order) and read the storer's
//      counter one sub-tile ahead of needing it.                                                                                                          0.384 ms (0.367 at hrx_recommended_pitches)
template <int D>
struct LdsQuadSink {
    static constexpr bool kSidq = true;
    uint32_t wbase, buf_stride;         // LDS: this lane's 16 rows of this walker's def in sub-tile buffer 0; the buffers' distance
    uint32_t sub, bi, nbuf;             // sub-tiles this walker has filled (four per tile, counted over all its groups); the buffer sub-tile `sub` goes to; the buffers (2 or 3)
    uint32_t filled_off, drained_off;   // LDS counters: this walker's; the storer's
    uint32_t seen;                      // the storer's counter as read right after the last sub-tile was posted: by the next sub-tile's first quad it has arrived with the walk's own reads
    uint32_t dbg;                       // (ablation build: kDbgSplitNoWalk — no sub-tile writes)
    unsigned char *rp;                  // (interface of GlobalSink: unused)
    __device__ __forceinline__ void quad(const int, const int p, const bool, const int, const uint4 &v) {
        const uint32_t j = (uint32_t)p >> 2;
        if ((j & 3u) == 0u && sub >= nbuf && (int32_t)(seen - (sub - nbuf + 1u)) < 0) ring_wait(drained_off, sub - nbuf + 1u);     // the buffer's previous sub-tile is out
        if (!(dbg & kDbgSplitNoWalk))
            *(volatile __attribute__((address_space(3))) v4u32 *)(uintptr_t)(wbase + bi * buf_stride + (j & 3u) * 16u) = v4u32{v.x, v.y, v.z, v.w};
        if ((j & 3u) == 3u) {
            ring_post_lds(filled_off, ++sub);
            bi = bi + 1u == nbuf ? 0u : bi + 1u;
            seen = lds_vol_u32(drained_off);
        }
    }
    __device__ __forceinline__ void row(const int) {}
};

constexpr uint32_t kSumBytes = 64u * 80u;     // per lane: st (8 B), en1 (8 B), 64 substr-id bytes
constexpr uint32_t kPmdPiece = 64u * 32u;     // per lane: dead, err_pos, err_state, err_char, acc_state

// CW: the CLASS-WIDE tables (hrx_lane.h) — a whole config of 4 .. 8 defs in one launch, one group per workgroup: D walkers, a combiner wave (FIN), a loader — 6 .. 10 waves: each def's walker looks its bytes' columns up in the def's 256-byte class LUT (four ds_read_u8 per quad,
// off the chain, before the tile's walk) and walks 256-byte rows; everything else is the D = 2, 3 kernel.  Batches of any size: the buffers' blocks of 65536 strings are addressed per group.
// (cfg 4 — three defs, two groups per workgroup — on these tables with a combiner wave: 3.03-3.23 ms against 2.89-2.94 for the WIDE kernel below; not taken.)
// FIN: the combiner is a wave of its own that walks nothing — W = D + 1 walker-like waves per group, all D walkers publish, the last wave merges, runs the reveal mask and stores the masked
// rows: with the last def's walk on top of D - 1 merges the combiner was the slowest wave of every group and set the launch's pace (0.54-0.65 of peak at 4 .. 7 defs whatever D).
template <int D, bool CW, bool FIN, bool SMO = false>
__global__ __launch_bounds__(FIN && (D >= 7 || (!CW && D >= 3)) ? 640 : 512) void witness_pmd_kernel(const WitnessArgs a, const uint32_t nring) {
    static_assert(!SMO || (CW && FIN), "string-major outputs: the CLASS-WIDE kernel with a combiner wave");
    // PA: the WIDE kernel with a combiner wave (cfg 4: three defs, two groups per workgroup, ten waves) has no LDS left for the walkers' 2-KiB status pieces — 75 KiB of tables,
    // two rings, six walkers' summaries: a walker's piece lies in its first summary slot, written when the combiner has read the group's last summary (pmd_fin_group_bytes)
    constexpr bool PA = FIN && !CW;
    constexpr int kMergeUnroll = D >= 7 ? 1 : 8;      // (the combiner's merge loop: rolled at seven and eight defs, see there)
    constexpr uint32_t kSubS = 16u * D + 4u;              // SMO: dwords per string of a sub-tile buffer
    constexpr uint32_t kSubBytes = 64u * kSubS * 4u;
    constexpr int RS = CW ? kCwRowShift : kWideRowShift;
    constexpr uint32_t W = FIN ? D + 1u : D;              // walker-like waves per group: the defs' walkers, the last one (FIN: an extra one) combining
    constexpr uint32_t kRowField = CW ? 0x3ffu : 0xffu, kRowMaskT = kRowField << RS;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t G = (blockDim.x >> 6) / (W + 1u + (SMO ? 1u : 0u));      // groups a workgroup walks at a time (SMO: + a storer wave per group, behind the loaders)
    // Roles in wave order: group 0's D walkers, group 1's, then the loaders.  The waves of a workgroup go to the CU's four SIMDs round-robin, so at D = 3, G = 2
    // group 0's combiner shares its SIMD with a loader and group 1's with another walker: group 0 of EVERY workgroup finishes at ~2435 us, group 1 at ~2918 us of a
    // 2957-us launch over 32768 rows (tools/front_width.py) — and that is the faster arrangement.  Round 5 dealt the roles so that both groups get the same company
    // (both finish together, the chip-wide write front 4-18 tiles wide instead of 110): 11 % SLOWER in a same-lease A/B (3.245 against 2.914 ms), and a gate that holds
    // loaders back once they are W tiles ahead of the chip's average moved the launch by -2 .. +2 %: profiles/r05_probes/cfg4_front_width.txt.
    const bool is_walker = wave < G * W;
    const bool is_storer = SMO && wave >= G * (W + 1u);
    const uint32_t lg = is_walker ? wave / W : is_storer ? wave - G * (W + 1u) : wave - G * W;
    const uint32_t d = is_walker ? wave % W : 0u;         // the def this walker walks (FIN: d == D is the combiner, which walks none)
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();

    // LDS per group: ring | (D - 1) x (2 summaries + piece) | counters: ready, freed[D], per publishing walker sum_prod, sum_cons, piece_prod; merged
    const uint32_t walker_bytes = 2u * kSumBytes + (PA ? 0u : kPmdPiece), piece_at = PA ? 0u : 2u * kSumBytes;
    const uint32_t group_bytes = nring * kPmTileBytes + (W - 1u) * walker_bytes + 192u + (SMO ? a.sm_bufs * kSubBytes + 8192u : 0u);  // the combiner publishes nothing (hrx_kernel.hpp pmd_group_bytes)
    const uint32_t ring_base = a.table_bytes + lg * group_bytes;
    const uint32_t wbase = ring_base + nring * kPmTileBytes;                  // walker areas of this group
    const uint32_t cnt = wbase + (W - 1u) * walker_bytes;
    const uint32_t ready_off = cnt, freed0 = cnt + 4u;                        // freed0 + 4 d
    auto sum_prod_off = [&](uint32_t dd) { return cnt + 48u + 12u * dd; };    // + 4: sum_cons, + 8: piece_prod  (W <= 9: freed[] ends at 40, these at 144)
    const uint32_t merged_off = cnt + 148u;
    const uint32_t filled0 = cnt + 152u, drained_off = cnt + 188u;            // SMO: filled0 + 4 d per walker, the storer's counter
    const uint32_t sub_base = cnt + 192u, mbuf = sub_base + a.sm_bufs * kSubBytes;  // SMO: the two sub-tile buffers, the masked rows' 8-KiB transpose buffer
    {
        const uint4 *src = CW ? reinterpret_cast<const uint4 *>(a.cw_image) : reinterpret_cast<const uint4 *>(a.wide_image);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = threadIdx.x; i < a.table_bytes / 16u; i += blockDim.x) dst[i] = src[i];
        if (!is_walker && !is_storer && lane < 48u) lds_store_u32(cnt + 4u * lane, 0);
    }
    __syncthreads();

    const uint32_t M = a.M, B = a.B;
    const uint32_t ntiles = (M + 63u) >> 6;
    const uint32_t g_first = blockIdx.x * G + lg, g_stride = gridDim.x * G;
    uint32_t seq = 0;

    if constexpr (SMO) {
        if (is_storer) {
            // ================================ storer ================================  (SMO: LDS sub-tiles -> the caller's string-major records)
            uint32_t sseq = 0, sbi = 0;     // sub-tiles drained so far; the buffer the next one is in
            const uint32_t run_bytes = a.rec_pitch * (uint32_t)D * 4u;      // a string's records (< 2^32 / 64: hrx_api.cpp checks the pitch)
            for (uint32_t g = g_first; g < a.n_groups; g += g_stride) {
                const uint32_t b0 = g * 64u;
                for (uint32_t t = 0; t < ntiles; ++t) {
#pragma unroll 1
                    for (uint32_t sub = 0; sub < 4u; ++sub, ++sseq) {
                        for (;;) {      // every walker has filled this sub-tile: the D counters in one round trip
                            bool ok = true;
#pragma unroll
                            for (uint32_t dd = 0; dd < (uint32_t)D; ++dd) ok &= (int32_t)(lds_vol_u32(filled0 + 4u * dd) - (sseq + 1u)) >= 0;
                            if (ok) break;
                            __builtin_amdgcn_s_sleep(1);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        const uint32_t row0 = t * 64u + sub * 16u;
                        const uint32_t buf = sub_base + sbi * kSubBytes;
                        sbi = sbi + 1u == a.sm_bufs ? 0u : sbi + 1u;
                        if (row0 < M) {     // (M % 16 == 0: a sub-tile lies below M entirely or not at all)
                            uint32_t w[4u * D][4];
#pragma unroll
                            for (uint32_t k = 0; k < 4u * D; ++k) {      // piece u of the sub-tile's 256 D: string u / 4 D, bytes 16 (u % 4 D) .. of its run = dwords q = 4 (u % 4 D) + i: row q / D, def q % D
                                const uint32_t u = k * 64u + lane, sidx = u / (4u * D), q0 = (u % (4u * D)) * 4u;
#pragma unroll
                                for (uint32_t i = 0; i < 4u; ++i) w[k][i] = lds_u32(buf + sidx * kSubS * 4u + ((q0 + i) % D) * 64u + ((q0 + i) / D) * 4u);
                            }
                            const unsigned char *base = reinterpret_cast<const unsigned char *>(a.records) + ((size_t)b0 * a.rec_pitch + ((a.debug & kDbgFixedLines) ? 0u : row0)) * D * 4u;   // (ablation: every sub-tile onto the strings' first rows)
#pragma unroll
                            for (uint32_t k = 0; k < 4u * D; ++k) {
                                const uint32_t u = k * 64u + lane, sidx = u / (4u * D), piece = u % (4u * D);
                                if (b0 + sidx < B && !(a.debug & kDbgSkipRecords)) store16_nt_so(base, sidx * run_bytes + piece * 16u, make_uint4(w[k][0], w[k][1], w[k][2], w[k][3]));
                            }
                        }
                        ring_post_lds(drained_off, sseq + 1u);
                    }
                }
            }
            return;
        }
    }
    if (!is_walker) {
        // ================================ loader ================================  (one per group, D consumers)
        constexpr uint32_t RT = 8u;
        const bool in_pm = (a.layout & 2u) != 0;
        const uint32_t my_groups = g_first < a.n_groups ? (a.n_groups - g_first + g_stride - 1u) / g_stride : 0u;
        const uint32_t total = my_groups * ntiles;
        const uint32_t row_cap = (uint32_t)a.stride - 16u;
        uint4 buf[RT * 4u];
        auto issue = [&](const uint32_t q, const uint32_t r) {
            const uint32_t g = g_first + (q / ntiles) * g_stride, t = q % ntiles;
            const uint32_t bl = min(g * 64u + lane, B - 1u);
            const uint32_t blk0 = (g * 64u / kPmBlock) * kPmBlock, nb = min(kPmBlock, B - blk0);     // the position-major buffers' block of 65536 strings this group lies in
            const size_t cmul = in_pm ? (size_t)nb : (size_t)1;
            const uint8_t *cptr = in_pm ? a.chars + (size_t)blk0 * a.stride + (size_t)(bl - blk0) * 16u : a.chars + (size_t)bl * a.stride;
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) buf[r * 4u + i] = *reinterpret_cast<const uint4 *>(cptr + (size_t)min(t * 64u + 16u * i, row_cap) * cmul);
        };
#pragma unroll
        for (uint32_t r = 0; r < RT; ++r)
            if (r < total) issue(r, r);
        for (uint32_t s0 = 0; s0 < total; s0 += RT) {
#pragma unroll
            for (uint32_t r = 0; r < RT; ++r) {
                const uint32_t sq = s0 + r;
                if (sq < total) {
                    if (sq >= nring) {
#pragma unroll
                        for (uint32_t dd = 0; dd < W; ++dd) ring_wait(freed0 + 4u * dd, sq - nring + 1u);   // every walker (and the combiner) is done with this slot
                    }
                    const uint32_t slot = ring_base + (sq % nring) * kPmTileBytes;
                    if (sq + RT <= total) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");   // RT-1 younger tiles x 4 loads may be in flight
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (uint32_t i = 0; i < 4u; ++i) {
                        uint4 v = buf[r * 4u + i];
                        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
                        *(__attribute__((address_space(3))) v4u32 *)(uintptr_t)(slot + i * 1024u + lane * 16u) = v4u32{v.x, v.y, v.z, v.w};
                    }
                    ring_post_lds(ready_off, sq + 1u);
                    if (sq + RT < total) issue(sq + RT, r);
                }
            }
        }
        return;
    }

    // ================================ walkers ================================
    WitnessArgs ad = a;           // the single-def view the D = 1 walk sees: its def's constants in slot 0
    ad.dc[0] = a.dc[min(d, (uint32_t)D - 1u)];
    const bool combiner = d == W - 1u;
    const uint32_t my_area = wbase + d * walker_bytes;
    uint32_t gi = 0, ready_seen = 0;     // the loader's counter as read at the end of the previous tile
    LdsQuadSink<D> lsink{sub_base + lane * kSubS * 4u + min(d, (uint32_t)D - 1u) * 64u, kSubBytes, 0u, 0u, a.sm_bufs, filled0 + 4u * min(d, (uint32_t)D - 1u), drained_off, 0u, a.debug, nullptr};
    for (uint32_t g = g_first; g < a.n_groups; g += g_stride, ++gi) {
        const uint32_t b0 = g * 64u;
        const uint32_t b = b0 + lane;
        const bool active = b < B;
        const uint32_t n_raw = a.lens[min(b, B - 1u)];   // lanes beyond the batch: exact shadows of string B - 1 (hrx_kernel_pm.hip), stores unpredicated
        const bool badlen = n_raw > M;
        const uint32_t n = badlen ? M : n_raw;
        const uint32_t min_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_min_u32(n));
        LaneRegs<1> L;
        L.e[0] = ad.dc[0].first_entry;  // states[d][0] = first_state_val: lib.rs:807
        L.mx[0] = 0;
        L.sid_prev = 0;
        L.ov_row = 0xffffffffu;
        uint32_t dead = 0, err_pos = 0, err_state = 0, err_char = 0;
        uint32_t acc_state[1] = {ad.dc[0].first_state};   // n == 0
        const uint32_t bc = active ? b : B - 1u;
        const uint32_t blk0 = (b0 / kPmBlock) * kPmBlock, nb = min(kPmBlock, B - blk0);                  // this group's block of the position-major buffers
        const size_t q4 = (M + 3u) / 4u, q8 = (M + 7u) / 8u;
        // this def's plane of the block's [M/4][RD][nb][4]: a pass of a multi-pass config (CW groups of up to eight defs, hrx_defs.hpp) writes planes rec_d0 .. of the caller's rec_D
        const uint32_t RD = a.rec_D ? a.rec_D : (uint32_t)D;
        const bool planes = a.rec_planes[0] != nullptr;      // ... or a buffer of its own per def (WitnessArgs::rec_planes): the D = 1 layout [M/4][nb][4] of the block
        unsigned char *rp = planes ? a.rec_planes[min(d, (uint32_t)D - 1u)] + ((size_t)blk0 * q4 + (bc - blk0)) * 16u
                                   : reinterpret_cast<unsigned char *>(a.records) + ((size_t)blk0 * q4 * RD + (size_t)(a.rec_d0 + min(d, (uint32_t)D - 1u)) * nb + (bc - blk0)) * 16u;
        const size_t rstep = planes ? (size_t)nb * 16u : (size_t)nb * 16u * RD;
        const size_t poff1[1] = {0};
        unsigned char *mp = reinterpret_cast<unsigned char *>(a.masked) + ((size_t)blk0 * q8 + (bc - blk0)) * 16u;
        const size_t mstep = (size_t)nb * 16u;
        const uint32_t lut = CW ? a.cw_lut_off + 256u * min(d, (uint32_t)D - 1u) : 0u;     // this def's class LUT
        // combiner state
        MaskCarry mc = {0, 0, 0, 0};
        uint32_t sum_prev = 0, ov_row = 0xffffffffu;
        const uint4 no_pend[8] = {};
        const bool nt_rec = !(a.debug & (kDbgNoNtStores | kDbgNoNtRecords)), nt_msk = !(a.debug & (kDbgNoNtStores | kDbgNoNtMasked));   // streaming stores (hrx_walk_pm.h)

        for (uint32_t t = 0; t < ntiles; ++t, ++seq) {
            const uint32_t t0 = t << 6;
            const uint32_t slot = ring_base + (seq % nring) * kPmTileBytes;
#ifdef HRX_STAMPS   // tools/front_width.py: when does this group's combiner reach each eighth of its rows (100-MHz wall clock, the same on every CU)
            if (a.stamps && combiner && lane == 0u && gi == 0u && (t % max(ntiles >> 3, 1u)) == 0u && t / max(ntiles >> 3, 1u) < 8u)
                a.stamps[(size_t)(blockIdx.x * G + lg) * 16u + t / max(ntiles >> 3, 1u)] = wall_clock64();
#endif
            ring_wait_seen(ready_off, seq + 1u, ready_seen);
            const uint32_t cons_seen = (!combiner && seq >= 2u) ? lds_vol_u32(sum_prod_off(d) + 4u) : 0u;    // (looked at again when the tile is walked: it arrives with the tile's bytes)
            uint4 cq[4];
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) cq[i] = lds_u128(slot + i * 1024u + lane * 16u);
            const uint32_t e_start = L.e[0];
            uint32_t sidq[16];
            TileBits tb;
            const bool full = (t0 + 64u < min_n);
            uint32_t tile_ov = 0, hb = 0;
            GlobalSink<1> sink{rp, poff1, rstep, rstep, 0, !(a.debug & kDbgSkipRecords), nt_rec, false, no_pend, mp, mstep, false, {}};
            const uint32_t cwl[16] = {cq[0].x, cq[0].y, cq[0].z, cq[0].w, cq[1].x, cq[1].y, cq[1].z, cq[1].w,
                                      cq[2].x, cq[2].y, cq[2].z, cq[2].w, cq[3].x, cq[3].y, cq[3].z, cq[3].w};
            uint4 ccol[4] = {};  // CW: the bytes' columns (class x 8) of this def, packed like the bytes
            if (FIN && combiner) {          // the combiner of the FIN variant walks nothing: its own share of the tile is empty
                tb = TileBits{0, 0, 0};
#pragma unroll
                for (int q = 0; q < 16; ++q) sidq[q] = 0;
            } else {
            if constexpr (CW) {
                uint32_t cw2[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const uint32_t x0 = lds_u8(lut + (cwl[q] & 0xffu)), x1 = lds_u8(lut + ((cwl[q] >> 8) & 0xffu)), x2 = lds_u8(lut + ((cwl[q] >> 16) & 0xffu)), x3 = lds_u8(lut + (cwl[q] >> 24));
                    cw2[q] = x0 | x1 << 8 | x2 << 16 | x3 << 24;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) ccol[i] = make_uint4(cw2[4 * i], cw2[4 * i + 1], cw2[4 * i + 2], cw2[4 * i + 3]);
            }
            const uint4 (&cwalk)[4] = CW ? ccol : cq;
            if (SMO && full) {
                tb = walk_tile_pm_wide<1, true, LdsQuadSink<D>, RS, CW>(L, cwalk, ad, lsink, 0, 0, tile_ov, sidq, acc_state);
            } else if (SMO) {
                tb = walk_tile_pm_wide<1, false, LdsQuadSink<D>, RS, CW>(L, cwalk, ad, lsink, (int)n - (int)t0, (int)M - 1 - (int)t0, tile_ov, sidq, acc_state);
            } else if (full) {
                tb = walk_tile_pm_wide<1, true, GlobalSink<1>, RS, CW>(L, cwalk, ad, sink, 0, 0, tile_ov, sidq, acc_state);
                if constexpr (!CW) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) hb |= cwl[q];
                    hb &= 0x80808080u;
                }
            } else {
                tb = walk_tile_pm_wide<1, false, GlobalSink<1>, RS, CW>(L, cwalk, ad, sink, (int)n - (int)t0, (int)M - 1 - (int)t0, tile_ov, sidq, acc_state);
                const uint32_t live_rows = n > t0 ? min(n - t0, 64u) : 0u;
#pragma unroll
                for (int q = 0; q < 16; ++q) {   // bytes at or beyond the string's length are not trusted
                    const uint32_t nby = live_rows > 4u * q ? min(live_rows - 4u * q, 4u) : 0u;
                    hb |= cwl[q] & (nby >= 4u ? 0xffffffffu : ((1u << (8u * nby)) - 1u));
                }
                hb &= CW ? 0u : 0x80808080u;      // (CW: every byte value has a column)
            }
            rp = sink.rp;
            // ---------------- undefined transition (lib.rs:817): rare slow path, re-walk the tile out of the ring slot ----------------
            if (__any(!dead && ((L.mx[0] & kRowMaskT) == ad.dc[0].dead_entry || hb != 0))) {
                if (!dead && ((L.mx[0] & kRowMaskT) == ad.dc[0].dead_entry || hb != 0)) {
                    uint32_t e = e_start;
                    const uint32_t live_rows = n > t0 ? min(n - t0, 64u) : 0u;
                    for (uint32_t p = 0; p < live_rows; ++p) {
                        const uint32_t c = smem[slot + (p >> 4) * 1024u + lane * 16u + (p & 15u)];
                        const uint32_t nx = CW ? lds_u32((e & kRowMaskT) | lds_u8(lut + c)) : c < 128u ? lds_u32((e & kRowMaskT) | (c << 3)) : ad.dc[0].dead_entry;
                        if ((nx & kRowMaskT) == ad.dc[0].dead_entry) {
                            err_pos = t0 + p;
                            err_state = ((e >> RS) & kRowField) - ad.dc[0].row_base;
                            err_char = c;
                            dead = 1;
                            break;
                        }
                        e = nx;
                    }
                }
            }
            // ---------------- accept state when n == M: row n does not exist, s[n] is the live state ----------------
            if (!full && n == t0 + 64u && t + 1 == ntiles) acc_state[0] = ((L.e[0] >> RS) & kRowField) - ad.dc[0].row_base;
            }   // (walks)
            ring_post_lds(freed0 + 4u * d, seq + 1u);   // done with the slot's bytes (the combiner keeps cq in registers)

            if (!combiner) {
                // ---- publish this def's share of the tile: start / end bitvectors and the byte-per-row substr ids
                const uint32_t sa = my_area + (seq & 1u) * kSumBytes + lane * 80u;
                if (seq >= 2u) ring_wait_seen(sum_prod_off(d) + 4u, seq - 1u, cons_seen);   // the combiner has read the summary that used this slot
                *(__attribute__((address_space(3))) v4u32 *)(uintptr_t)sa = v4u32{(uint32_t)tb.st, (uint32_t)(tb.st >> 32), (uint32_t)tb.en1, (uint32_t)(tb.en1 >> 32)};
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i)
                    *(__attribute__((address_space(3))) v4u32 *)(uintptr_t)(sa + 16u + 16u * i) = v4u32{sidq[4 * i], sidq[4 * i + 1], sidq[4 * i + 2], sidq[4 * i + 3]};
                ring_post_lds(sum_prod_off(d), seq + 1u);
                ready_seen = lds_vol_u32(ready_off);
                continue;
            }
            // ================= combiner (the last def's walker): sums over the defs, reveal mask, masked rows =================
            uint64_t st = tb.st, en1 = tb.en1, ov_st = 0, ov_en = 0;
            // (the merge loop below, seven and eight defs: rolled — unrolled, the summaries' loads of all iterations are in flight at once and the nine / ten waves' 168 VGPRs spill)
            for (;;) {      // every walker has published this tile: the counters in one round trip
                bool ok = true;
#pragma unroll
                for (uint32_t dd = 0; dd + 1u < W; ++dd) ok &= (int32_t)(lds_vol_u32(sum_prod_off(dd)) - (seq + 1u)) >= 0;
                if (ok) break;
                __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll kMergeUnroll
            for (uint32_t dd = 0; dd + 1u < W; ++dd) {
                const uint32_t sa = wbase + dd * walker_bytes + (seq & 1u) * kSumBytes + lane * 80u;
                const uint4 h = lds_u128(sa);
                const uint64_t ost = (uint64_t)h.x | ((uint64_t)h.y << 32), oen = (uint64_t)h.z | ((uint64_t)h.w << 32);
                ov_st |= st & ost;   // two defs raise is_start on the same row: out of contract
                ov_en |= en1 & oen;
                st |= ost;
                en1 |= oen;
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i) {
                    const uint4 v = lds_u128(sa + 16u + 16u * i);
                    sidq[4 * i] += v.x; sidq[4 * i + 1] += v.y; sidq[4 * i + 2] += v.z; sidq[4 * i + 3] += v.w;   // byte sums <= 255 (finalize_defs)
                }
                ring_post_lds(sum_prod_off(dd) + 4u, seq + 1u);
            }
            if (ov_row == 0xffffffffu) {
                if (ov_st) ov_row = t0 + (uint32_t)ctz64(ov_st);
                if (ov_en) ov_row = min(ov_row, t0 + (uint32_t)ctz64(ov_en) + 1u);
            }
            if (a.summary) {   // a pass of a multi-pass config: this group's share of the tile for the combine launch (hrx_kernel_mp.hip) — [tile][5][B][16 B], as the finisher of hrx_kernel_pm.hip writes it
                if (active) {
                    uint4 *sp = reinterpret_cast<uint4 *>(a.summary) + ((size_t)t * 5u * B + b);
                    sp[0] = make_uint4((uint32_t)st, (uint32_t)(st >> 32), (uint32_t)en1, (uint32_t)(en1 >> 32));
#pragma unroll
                    for (uint32_t i = 0; i < 4u; ++i) sp[(size_t)(i + 1u) * B] = make_uint4(sidq[4 * i], sidq[4 * i + 1], sidq[4 * i + 2], sidq[4 * i + 3]);
                }
                continue;
            }
            // id-changed bits: byte p of the sums against byte p - 1 (the previous tile's last byte for p = 0)
            uint64_t ch = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const uint32_t x = sidq[q], y = (x << 8) | (q ? (sidq[q - 1] >> 24) : sum_prev);
                const uint32_t dxy = x ^ y;
                const uint32_t nz = ((dxy | ((dxy & 0x7f7f7f7fu) + 0x7f7f7f7fu)) >> 7) & 0x01010101u;   // 1 per non-zero byte
                ch |= (uint64_t)(((nz * 0x01020408u) >> 24) & 0xfu) << (4 * q);                       // byte i -> bit i
            }
            sum_prev = sidq[15] >> 24;
            TileBits all{st, en1, ch};
            TileMasks tm = tile_masks<64>(all, mc, t0, tile_is_exact(t0, n, M), rows_below(t0, n));
            if (!active) tm.fix = 0;
            uint64_t fixm = __ballot(tm.fix != 0);
            if (a.debug & kDbgSkipFixups) fixm = 0;
            while (fixm) {   // an earlier optimistic end_mask = 1 turned out wrong: zero those masked rows (rare)
                const int j = __ffsll((unsigned long long)fixm) - 1;
                fixm &= fixm - 1;
                const uint32_t fs = (uint32_t)__builtin_amdgcn_readlane((int)tm.fix_start, j);
                const uint32_t bj = b0 + (uint32_t)j;
                for (uint32_t r = fs + lane; r < t0; r += 64u) {
                    if (SMO) a.masked[(size_t)bj * a.msk_pitch + r] = 0;
                    else a.masked[((size_t)blk0 * q8 + (size_t)(r >> 3) * nb + (bj - blk0)) * 8u + (r & 7u)] = 0;     // (in this group's block of the buffer)
                }
            }
            {
                // (streamed unless a string of the wave has an open optimistic span: rows that may be zeroed later stay in L2 for the repair — hrx_kernel_pm.hip octets_out)
                const bool nt_tile = nt_msk && !(!(a.nt_mix & kNtMixNoOpenSpan) && __any(mc.pend != 0u));
                const uint32_t mlo = (uint32_t)tm.mask, mhi = (uint32_t)(tm.mask >> 32);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const uint32_t mbyte = ((q < 4 ? mlo : mhi) >> (8 * (q & 3))) & 0xffu;
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (mbyte) v = masked_octet(cwl[2 * q], cwl[2 * q + 1], sidq[2 * q], sidq[2 * q + 1], mbyte);  // lib.rs:752-761
                    if constexpr (SMO) {   // string-major masked rows [B][pitch]: a string's 64 rows are one 128-byte line — through LDS, so that a store instruction writes the full lines of eight strings
                        *(__attribute__((address_space(3))) v4u32 *)(uintptr_t)(mbuf + (lane * 8u + ((uint32_t)q ^ (lane & 7u))) * 16u) = v4u32{v.x, v.y, v.z, v.w};
                    } else {
                        if (t0 + (uint32_t)q * 8u < M && !(a.debug & kDbgSkipMasked)) store16(mp + (size_t)q * mstep, v, nt_tile);
                    }
                }
                if constexpr (SMO) {
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const uint32_t js0 = lane >> 3, w8 = lane & 7u;
#pragma unroll
                    for (uint32_t it = 0; it < 8u; ++it) {
                        const uint32_t js = it * 8u + js0;
                        const uint4 v = lds_u128(mbuf + (js * 8u + (w8 ^ (js & 7u))) * 16u);
                        if (b0 + js < B && t0 + w8 * 8u < M && !(a.debug & kDbgSkipMasked))
                            store16(reinterpret_cast<unsigned char *>(a.masked) + ((size_t)(b0 + js) * a.msk_pitch + t0) * 2u + (size_t)w8 * 16u, v, nt_tile);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                mp += 8u * mstep;
            }
        }
#ifdef HRX_STAMPS
        if (a.stamps && combiner && lane == 0u && gi == 0u) a.stamps[(size_t)(blockIdx.x * G + lg) * 16u + 8u] = wall_clock64();
#endif
        // ---------------- per-string status: every def's walker publishes its piece, the combiner merges ----------------
        const uint32_t pa = my_area + piece_at + lane * 32u;
        if (!combiner) {
            if (PA) ring_wait(sum_prod_off(d) + 4u, seq);      // the combiner has read every summary of this group: the slot is free for the piece
            *(__attribute__((address_space(3))) v4u32 *)(uintptr_t)pa = v4u32{dead, err_pos, err_state, err_char};
            *(__attribute__((address_space(3))) v4u32 *)(uintptr_t)(pa + 16u) = v4u32{acc_state[0], 0u, 0u, 0u};
            ring_post_lds(sum_prod_off(d) + 8u, gi + 1u);
            // wait until the combiner has merged before the next group's piece overwrites this one
            ring_wait(merged_off, gi + 1u);
            continue;
        }
        uint32_t m_dead = 0, accept = 0;
        uint32_t m_def = 0, m_pos = 0, m_state = 0, m_char = 0;     // the LOWEST def's undefined transition: the reference walks the defs in order (lib.rs:806)
#pragma unroll
        for (uint32_t dd = 0; dd < (uint32_t)D; ++dd) {
            uint32_t w_dead, w_pos, w_state, w_char, w_acc;
            if (dd + 1u < W) {
                ring_wait(sum_prod_off(dd) + 8u, gi + 1u);
                const uint32_t oa = wbase + dd * walker_bytes + piece_at + lane * 32u;
                const uint4 x0 = lds_u128(oa), x1 = lds_u128(oa + 16u);
                w_dead = x0.x; w_pos = x0.y; w_state = x0.z; w_char = x0.w; w_acc = x1.x;
            } else {
                w_dead = dead; w_pos = err_pos; w_state = err_state; w_char = err_char; w_acc = acc_state[0];
            }
            if ((w_dead & 1u) && !m_dead) { m_def = dd; m_pos = w_pos; m_state = w_state; m_char = w_char; }
            m_dead |= w_dead & 1u;
            accept |= (w_acc == a.dc[dd].accepted_state ? 1u : 0u) << dd;
        }
        ring_post_lds(merged_off, gi + 1u);
        if (active) {
            uint64_t sw;
            if (badlen) sw = kStatusBadLength;
            else if (m_dead) {
                sw = status_invalid(m_def, m_pos, m_state, m_char);
            } else if (ov_row != 0xffffffffu) sw = status_overlap(ov_row);
            else sw = status_ok(accept);
            a.status[b] = sw;
        }
    }
}

template <int D, bool CW, bool FIN, bool SMO = false>
static hipError_t launch_pmd(const WitnessArgs &a, const LaunchInfo &li, hipStream_t stream) {
    auto kern = witness_pmd_kernel<D, CW, FIN, SMO>;
    static std::atomic<size_t> granted[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipError_t e = ensure_lds(kern, granted[dev & 63], li.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(li.grid), dim3(64 * li.waves_per_wg), li.lds_bytes, stream, a, (uint32_t)li.nslots);
    return hipGetLastError();
}

hipError_t launch_witness_pmd(const WitnessArgs &a, const LaunchInfo &li, hipStream_t stream) {
    if (a.cw_image && !(a.layout & 1u)) {   // ... with string-major outputs: + a storer wave (four and five defs)
        return a.D == 4 ? launch_pmd<4, true, true, true>(a, li, stream) : a.D == 5 ? launch_pmd<5, true, true, true>(a, li, stream) : hipErrorInvalidValue;
    }
    if (a.cw_image) {   // CLASS-WIDE tables: D walkers + a combiner wave + a loader per group
        switch (a.D) {
            case 4: return launch_pmd<4, true, true>(a, li, stream);
            case 5: return launch_pmd<5, true, true>(a, li, stream);
            case 6: return launch_pmd<6, true, true>(a, li, stream);
            case 7: return launch_pmd<7, true, true>(a, li, stream);
            case 8: return launch_pmd<8, true, true>(a, li, stream);
            default: return hipErrorInvalidValue;
        }
    }
    if (li.pmd_fin) return a.D == 2 ? launch_pmd<2, false, true>(a, li, stream) : a.D == 3 ? launch_pmd<3, false, true>(a, li, stream) : hipErrorInvalidValue;
    return a.D == 2 ? launch_pmd<2, false, false>(a, li, stream) : a.D == 3 ? launch_pmd<3, false, false>(a, li, stream) : hipErrorInvalidValue;
}

}  // namespace hrx
