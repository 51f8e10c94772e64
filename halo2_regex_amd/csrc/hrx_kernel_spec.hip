// hrx_kernel_spec.hip — CHUNKED walk for batches that leave most walker slots empty.
//
// derive_states is strictly sequential per string (src/lib.rs:808-819): a batch of fewer strings than the chip has lanes runs
// for as long as ONE string's dependent chain, n x ~50 ns (8192 x 32768-byte strings: 0.94 ms with the pair-step table, a
// quarter of the memory system's rate).  What lets a string be cut: a DFA forgets.  Started from ALL of its states on the same
// bytes, the walks merge within a few dozen bytes into a handful of survivors — for the reference's definitions two: the state
// every "ordinary" history leads to and the absorbing accept state behind a finished match (the case that defeats a plain
// warm-up from the start state: no warm-up reaches "already matched").  So:
//
//   scout    every (string, chunk, def): walk the chunk's first kSpecPrefix bytes from every real state, keep the (at most
//            kSpecSlots) distinct survivors and which survivor each start state became, then walk only the survivors to the
//            chunk's end; no rows are written — a chunk costs ~S x 32 + 2 x (chunk - 32) table lookups, several independent
//            chains per lane.
//   compose  a lane per string, one wave per (64 strings, def): from first_state, chunk by chunk, the state at every chunk's first row
//            (an evaluation of the chunk's scout row, loaded four chunks ahead of the chain) and the substr id / end flag of the
//            transition into it.  A chunk whose start states did not merge into kSpecSlots survivors is walked here by the whole
//            wave, a 64th of it per lane from every state — correct for any DFA, fast for forgetful ones.
//   walk     the ordinary loader / walker / finisher kernel (hrx_kernel_pm.hip) over chunks as virtual groups: B x C "strings"
//            fill the chip, the launch is bound by the memory system again.  Rows are final except for what crosses a
//            chunk's borders in the reveal-mask scans (lib.rs:598-714): a chunk assumes no open span at its first row
//            (start_mask carry 0) and, as everywhere, end_mask = 1 for rows after its last backward event.
//   stitch   one thread per string: the true carries from the chunks' one-word summaries ("last event wins": a chunk with a
//            forward event fixes the carry out of it, one without passes it on; the first deciding tile of a later chunk
//            tells the end mask of the rows before it); a chunk whose assumption was wrong has its masked rows recomputed
//            from its finished records and the input bytes (rare with real definitions: a revealed substring must straddle a
//            chunk border); the chunks' status words merge (lowest chunk's undefined transition, lowest overlap row, the accept
//            state of the chunk that holds row n).
#include <hip/hip_runtime.h>

#include "hrx_device.h"

namespace hrx {

// The prefix in two stages: every real state over kSpecStageA bytes (the walks merge fast: 29 states -> at most 7 after 8 bytes of the
// reference's and the header definitions, profiles/r03_probes/convergence.txt) — the distinct states reached are the chunk's KEYS —
// then only the keys over the rest of the prefix, then the distinct survivors of that to the chunk's end.
constexpr uint32_t kSpecStageA = 8, kSpecPrefix = 32;
constexpr uint32_t kSpecKeys = 8;      // distinct states after stage A per (string, chunk, def); more: the compose launch walks that chunk itself
constexpr uint32_t kSpecSlots = 4;     // survivors walked to the chunk's end; more: likewise

// ---------------------------------------------------------------------------------------------
// scout: lane = (chunk, string); the narrow fused table in LDS
// ---------------------------------------------------------------------------------------------
template <bool COMPACT>
__global__ __launch_bounds__(1024) void spec_scout_kernel(const SpecArgs a) {
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(COMPACT ? (const void *)a.cimage : (const void *)a.table_image);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = threadIdx.x; i < (COMPACT ? a.cimage_bytes : a.table_bytes) / 16u; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const uint32_t Bpad = a.n_groups * 64u;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t rows = a.tiles_per_chunk * 64u;
    const uint32_t row_cap = (uint32_t)a.stride - 16u;
    for (uint32_t vw_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); vw_ < a.n_groups * a.C; vw_ += gridDim.x * (blockDim.x >> 6)) {
        const uint32_t vw = (uint32_t)__builtin_amdgcn_readfirstlane((int)vw_);
        const uint32_t k = vw / a.n_groups, g = vw % a.n_groups;
        const uint32_t b = g * 64u + lane, bc = min(b, a.B - 1u);
        const uint32_t r0 = k * rows;
        const uint32_t n = min(a.lens[bc], a.M);
        // nothing of this chunk matters if no string of the wave reaches it (rows >= n are padding: lib.rs:404-418)
        if (__ballot(n > r0) == 0ull) continue;
        const uint32_t blk0 = (g * 64u / kPmBlock) * kPmBlock, nb = min(kPmBlock, a.B - blk0);
        // position-major input: 16-byte piece i of the string at + i * nb * 16; string-major: at + 16 i
        const uint8_t *cptr = a.in_pm ? a.chars + (size_t)blk0 * a.stride + (size_t)(bc - blk0) * 16u : a.chars + (size_t)bc * a.stride;
        const size_t cmul = a.in_pm ? (size_t)nb : (size_t)1;
        auto piece = [&](const uint32_t i) -> uint4 {   // bytes [r0 + 16 i, r0 + 16 i + 16) of the string (clamped inside its stride)
            return *reinterpret_cast<const uint4 *>(cptr + (size_t)min(r0 + 16u * i, row_cap) * cmul);
        };
        const uint4 p0 = piece(0), p1 = piece(1);
        const uint32_t pw[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
        for (uint32_t d = 0; d < a.D; ++d) {
            const uint32_t S = a.n_states[d], base = a.dc[d].row_base;
            // A chain word `e` names a state.  Narrow table: the table entry (next row << kNextShift | tags), one v_and_or + ds_read_b32 per step at a random one of 64
            // banks (column = byte).  COMPACT: the LDS byte address of the state's row in the class-indexed u16 table, one v_add + ds_read_u16 per step at one of the
            // row's few dwords, and per byte (shared by the chains of the def) one ds_read_u8 of the 256-byte class LUT, which lies in 64 different banks.
            const uint32_t c_lut = COMPACT ? a.c_lut[d] : 0u, c_tab = COMPACT ? a.c_tab[d] : 0u, c_rowb = COMPACT ? a.c_rowb[d] : 0u, c_inv = COMPACT ? a.c_inv[d] : 0u;
            auto e_of = [&](const uint32_t s) -> uint32_t { return COMPACT ? c_tab + s * c_rowb : (base + s) << kNextShift; };
            auto st_of = [&](const uint32_t e) -> uint32_t { return COMPACT ? ((e - c_tab) * c_inv) >> 16 : (e >> kNextShift) - base; };
            auto name_of = [&](const uint32_t e) -> uint32_t { return COMPACT ? e : e & ~kTagMask; };
            auto col_of = [&](const uint32_t byte) -> uint32_t { return COMPACT ? (uint32_t)*(__attribute__((address_space(3))) const uint8_t *)(uintptr_t)(c_lut + byte) : byte << 2; };
            auto step = [&](const uint32_t e, const uint32_t col) -> uint32_t {
                if (COMPACT) return (uint32_t)*(__attribute__((address_space(3))) const uint16_t *)(uintptr_t)(e + col);
                return lds_u32((e & ~kTagMask) | col);
            };
            uint32_t pcol[kSpecPrefix];      // the prefix bytes' columns (COMPACT: their classes, looked up once for all chains of this def)
#pragma unroll
            for (uint32_t i = 0; i < kSpecPrefix; ++i) pcol[i] = col_of((pw[i >> 2] >> (8u * (i & 3u))) & 0xffu);
            uint32_t key[kSpecKeys], NK = 0, fail = 0;
#pragma unroll
            for (uint32_t j = 0; j < kSpecKeys; ++j) key[j] = 0xffffffffu;
            // this (chunk, def, string)'s row of the scratch: smax bytes "which key did start state s become" + the 32-byte record below
            uint8_t *rowp = a.rows + ((size_t)(k * a.D + d) * Bpad + b) * a.row_bytes;
            // ---- stage A: every real state over the chunk's first kSpecStageA bytes, four independent chains at a time; cls[s] = the
            // state reached (chunk 0 starts in first_state: one candidate)
            const uint32_t only = k == 0u ? a.dc[d].first_state : 0xffffffffu;          // chunk 0: the one candidate
            const uint32_t s_begin = k == 0u ? (only & ~3u) : 0u, s_end = k == 0u ? only + 1u : S;
            for (uint32_t s0 = s_begin; s0 < s_end; s0 += 4u) {
                uint32_t e[4];
#pragma unroll
                for (uint32_t j = 0; j < 4u; ++j) e[j] = e_of(min(s0 + j, S - 1u));
#pragma unroll
                for (uint32_t i = 0; i < kSpecStageA; ++i) {
#pragma unroll
                    for (uint32_t j = 0; j < 4u; ++j) e[j] = step(e[j], pcol[i]);
                }
                uint32_t clsw = 0;        // the four candidates' states after stage A, one byte each
#pragma unroll
                for (uint32_t j = 0; j < 4u; ++j) {
                    if (s0 + j < s_end && (only == 0xffffffffu || s0 + j == only)) {
                        const uint32_t v = name_of(e[j]);
                        bool have = false;
#pragma unroll
                        for (uint32_t q = 0; q < kSpecKeys; ++q) have = have || key[q] == v;
                        if (!have) {
                            if (NK < kSpecKeys) {
#pragma unroll
                                for (uint32_t q = 0; q < kSpecKeys; ++q) if (q == NK) key[q] = v;
                                ++NK;
                            } else {
                                fail = 1;
                            }
                        }
                        clsw |= (st_of(v) & 0xffu) << (8u * j);
                    }
                }
                if (b < a.B) *reinterpret_cast<uint32_t *>(rowp + s0) = clsw;
            }
            // ---- stage B: the keys over the rest of the prefix (as many groups of four chains as the wave's busiest lane has keys), then
            // their distinct survivors: slot[] and, per key, which slot it became
            uint32_t NKw = NK;
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) NKw = max(NKw, (uint32_t)__shfl_xor((int)NKw, sft, 64));
            NKw = (uint32_t)__builtin_amdgcn_readfirstlane((int)NKw);
            uint32_t kb[kSpecKeys];
#pragma unroll
            for (uint32_t j = 0; j < kSpecKeys; ++j) kb[j] = key[j] == 0xffffffffu ? e_of(0u) : key[j];
#pragma unroll
            for (uint32_t h = 0; h < kSpecKeys; h += 4u) {
                if (h < NKw) {
#pragma unroll
                    for (uint32_t i = kSpecStageA; i < kSpecPrefix; ++i) {
#pragma unroll
                        for (uint32_t j = 0; j < 4u; ++j) kb[h + j] = step(kb[h + j], pcol[i]);
                    }
                }
            }
            uint32_t slot[kSpecSlots], kslot[kSpecKeys], K = 0;
#pragma unroll
            for (uint32_t j = 0; j < kSpecSlots; ++j) slot[j] = 0xffffffffu;
#pragma unroll
            for (uint32_t j = 0; j < kSpecKeys; ++j) {
                kslot[j] = 0;
                if (j < NK) {
                    const uint32_t v = name_of(kb[j]);
                    uint32_t idx = 0xffu;
#pragma unroll
                    for (uint32_t q = 0; q < kSpecSlots; ++q) if (idx == 0xffu && slot[q] == v) idx = q;
                    if (idx == 0xffu) {
                        if (K < kSpecSlots) {
#pragma unroll
                            for (uint32_t q = 0; q < kSpecSlots; ++q) if (q == K) slot[q] = v;
                            idx = K++;
                        } else {
                            fail = 1; idx = 0;
                        }
                    }
                    kslot[j] = idx;
                }
            }
            // ---- the survivors to the chunk's end.  A QUASI-ABSORBING survivor (hrx_kernel.hpp SpecArgs::qabs: every byte keeps it where it is
            // or — only bytes NO state has a transition for — kills it: the accept state behind a finished match) is not walked: it ends
            // where it started as long as no walked survivor died (a byte without any column kills every walked one; if one did die the
            // chunk is flagged below, the byte may have been undefined for that state only).  The walked ones go first; as many
            // chains as the wave's busiest lane has of them (at least one: somebody has to see such a byte).
            uint32_t walk[kSpecSlots], idle[kSpecSlots], nw = 0, ni = 0;      // slot numbers in walking order
#pragma unroll
            for (uint32_t j = 0; j < kSpecSlots; ++j) { walk[j] = 0; idle[j] = 0; }
#pragma unroll
            for (uint32_t j = 0; j < kSpecSlots; ++j) {
                if (j < K) {
                    const uint32_t stt = st_of(slot[j]);
                    const bool q = stt < S && ((a.qabs[d][stt >> 5] >> (stt & 31u)) & 1u);
#pragma unroll
                    for (uint32_t t = 0; t < kSpecSlots; ++t) {
                        if (q && t == ni) idle[t] = j;
                        if (!q && t == nw) walk[t] = j;
                    }
                    if (q) ++ni; else ++nw;
                }
            }
            if (nw == 0u && K > 0u) { walk[0] = idle[0]; nw = 1; }           // (slot idle[0] is then walked AND derived: the same result)
            uint32_t Kw = nw;
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) Kw = max(Kw, (uint32_t)__shfl_xor((int)Kw, sft, 64));
            Kw = (uint32_t)__builtin_amdgcn_readfirstlane((int)Kw);
            uint32_t e[kSpecSlots], em1[kSpecSlots];
#pragma unroll
            for (uint32_t j = 0; j < kSpecSlots; ++j) {
                uint32_t v = e_of(0u);
#pragma unroll
                for (uint32_t t = 0; t < kSpecSlots; ++t) if (j < nw && walk[j] == t && slot[t] != 0xffffffffu) v = slot[t];
                e[j] = v; em1[j] = v;
            }
            const uint32_t npieces = rows / 16u;
            // (one copy of the loop per chain count: a count in a register costs a branch per lookup)
            auto to_end = [&](auto kc) {
                constexpr uint32_t KC = decltype(kc)::value;
                // the input runs kAhead pieces (16 bytes each = 16 lookups of ~70 cycles) ahead of the chains: with two pieces ahead a load
                // had 0.9 us to arrive, less than a memory access takes while the chip is busy
                constexpr uint32_t kAhead = 4;
                const uint32_t p0 = kSpecPrefix / 16u;
                uint4 pc[kAhead];
#pragma unroll
                for (uint32_t u = 0; u < kAhead; ++u) pc[u] = piece(min(p0 + u, npieces - 1u));
                for (uint32_t i = p0; i < npieces; i += kAhead) {
#pragma unroll
                    for (uint32_t u = 0; u < kAhead; ++u) {
                        if (i + u < npieces) {
                            const uint32_t w[4] = {pc[u].x, pc[u].y, pc[u].z, pc[u].w};
                            pc[u] = piece(min(i + u + kAhead, npieces - 1u));
                            uint32_t col[16];
#pragma unroll
                            for (uint32_t q = 0; q < 16u; ++q) col[q] = col_of((w[q >> 2] >> (8u * (q & 3u))) & 0xffu);    // (COMPACT: 16 LUT reads off the chain)
#pragma unroll
                            for (uint32_t q = 0; q < 16u; ++q) {
                                if (q == 15u) {
#pragma unroll
                                    for (uint32_t j = 0; j < KC; ++j) em1[j] = e[j];    // the state BEFORE the piece's last byte (kept for the chunk's last piece)
                                }
#pragma unroll
                                for (uint32_t j = 0; j < KC; ++j) e[j] = step(e[j], col[q]);
                            }
                        }
                    }
                }
            };
            switch (Kw) {
                case 0: case 1: to_end(std::integral_constant<uint32_t, 1>{}); break;
                case 2: to_end(std::integral_constant<uint32_t, 2>{}); break;
                case 3: to_end(std::integral_constant<uint32_t, 3>{}); break;
                default: to_end(std::integral_constant<uint32_t, 4>{}); break;
            }
            if (b < a.B) {
                // per KEY (a state reached after stage A): the key itself, where it is at the chunk's end and before the chunk's last byte
                // one 32-byte record per (chunk, def, string): keys[8] | where each key ends [8] | where it is before the last byte [8] | fail | pad
                uint32_t rw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                const uint32_t dead_row = S + 1u;
                // did a walked survivor die, and had it died before the chunk's last byte already?
                bool died = false, died_m1 = false;
#pragma unroll
                for (uint32_t j = 0; j < kSpecSlots; ++j) {
                    if (j < nw) {
                        died = died || st_of(e[j]) == dead_row;
                        died_m1 = died_m1 || st_of(em1[j]) == dead_row;
                    }
                }
                // A walked survivor that died says nothing about the derived ones unless the byte that killed it has no column at all: in a
                // PARTIAL DFA a byte may be undefined for the walked state only, and the quasi-absorbing one lives on (fuzz seed 2499: two-byte
                // alphabet, 36 states, 60 of 200 strings reported dead at the next chunk's first row).  Such a chunk goes to the compose
                // wave's exact walk; strings of the reference's total DFAs get here only with a byte outside the alphabet, i.e. as errors.
                if (died && ni != 0u) fail = 1;
                uint32_t s_end_v[kSpecSlots], s_endm1_v[kSpecSlots];
#pragma unroll
                for (uint32_t t = 0; t < kSpecSlots; ++t) {          // slot t: walked as chain j, or derived
                    uint32_t end = st_of(slot[t]), endm1 = end;
                    bool walked = false;
#pragma unroll
                    for (uint32_t j = 0; j < kSpecSlots; ++j)
                        if (j < nw && walk[j] == t) { end = st_of(e[j]); endm1 = st_of(em1[j]); walked = true; }
                    if (!walked) { if (died) end = dead_row; if (died_m1) endm1 = dead_row; }
                    s_end_v[t] = end; s_endm1_v[t] = endm1;
                }
#pragma unroll
                for (uint32_t j = 0; j < kSpecKeys; ++j) {
                    uint32_t end = 0, endm1 = 0;
#pragma unroll
                    for (uint32_t t = 0; t < kSpecSlots; ++t) if (kslot[j] == t) { end = s_end_v[t]; endm1 = s_endm1_v[t]; }
                    const uint32_t kv = j < NK ? st_of(key[j]) & 0xffu : 0xffu;     // (0xff: no such key; states are <= 254 here)
                    rw[j >> 2] |= kv << (8u * (j & 3u));
                    rw[2u + (j >> 2)] |= (end & 0xffu) << (8u * (j & 3u));
                    rw[4u + (j >> 2)] |= (endm1 & 0xffu) << (8u * (j & 3u));
                }
                rw[6] = fail;
                uint4 *rp = reinterpret_cast<uint4 *>(rowp + a.smax);
                rp[0] = make_uint4(rw[0], rw[1], rw[2], rw[3]);
                rp[1] = make_uint4(rw[4], rw[5], rw[6], rw[7]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// compose: lane = string, one wave per (64 strings, def)
// ---------------------------------------------------------------------------------------------
// The chain over a string's chunks: s[k + 1] = "where does s[k] end in chunk k", read off chunk k's scout row (64 bytes for DFAs of up
// to 32 states: which key each start state became | keys | where each key ends).  A lane walks its own string; the rows do not depend
// on the state, so they are loaded kAheadRows chunks ahead of the chain, and the (state, next) -> tag lookups of the init words trail
// kAheadRows chunks behind it: no memory access waits on the chain.  (One wave per (string, def) with the chunks in its lanes paid
// 64 lanes of VALU work for one useful answer per step: 31 us at D = 1, 81 us at D = 3 for 8192 strings.)
// A chunk the scout gave up on is walked by the whole wave for the one lane that needs it (spec_walk_flagged below).
constexpr uint32_t kAheadRows = 4;

// chunk k of string b (def d): the wave walks it — lane l takes the chunk's l-th 64th from EVERY real state (four independent chains at
// a time, table and bytes from L2), its map "entered in s -> left in" goes to LDS; then the true state runs through the 64 maps.
// ~S x rows / 64 lookups per lane instead of a chain of `rows` dependent ones (200 us for 1024 rows).  All arguments wave-uniform.
__device__ __forceinline__ void spec_walk_flagged(const SpecArgs &a, const uint32_t b, const uint32_t d, const uint32_t k, uint32_t &s, uint32_t &prev, const uint32_t lane, uint8_t *maps) {
    const uint32_t S = a.n_states[d], base = a.dc[d].row_base, rows = a.tiles_per_chunk * 64u, r0 = k * rows;
    const uint32_t *T = a.table_image + (size_t)base * 256u;
    const uint32_t blk0 = (b / kPmBlock) * kPmBlock, nb = min(kPmBlock, a.B - blk0);
    const uint8_t *cptr = a.in_pm ? a.chars + (size_t)blk0 * a.stride + (size_t)(b - blk0) * 16u : a.chars + (size_t)b * a.stride;
    const size_t cmul = a.in_pm ? (size_t)nb * 16u : (size_t)16u;     // byte r of the string at + (r >> 4) * cmul + (r & 15)
    const uint32_t per = rows >> 6;
    for (uint32_t s0 = 0; s0 < S; s0 += 4u) {
        uint32_t e[4], em1[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) { e[j] = min(s0 + j, S - 1u); em1[j] = e[j]; }
        for (uint32_t i = 0; i < per; ++i) {
            const uint32_t r = r0 + lane * per + i;
            const uint32_t c = cptr[(size_t)(r >> 4) * cmul + (r & 15u)];
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) { em1[j] = e[j]; e[j] = (T[e[j] * 256u + c] >> kNextShift) - base; }    // (the dead row S + 1 leads to itself)
        }
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            if (s0 + j < S) {
                maps[lane * a.smax + s0 + j] = (uint8_t)e[j];
                if (lane == 63u) maps[64u * a.smax + s0 + j] = (uint8_t)em1[j];
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t l = 0; l < 64u && s < S; ++l) {
        if (l == 63u) prev = (uint32_t)__builtin_amdgcn_readfirstlane((int)maps[64u * a.smax + s]);
        s = (uint32_t)__builtin_amdgcn_readfirstlane((int)maps[l * a.smax + s]);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// SMALL: at most 32 table rows per def — a chunk's "which key did start state s become" bytes ride in two registers (a run-time test here
// put a wait for every outstanding load into each step)
template <bool SMALL>
__global__ __launch_bounds__(64) void spec_compose_kernel(const SpecArgs a) {
    if (blockIdx.x == 0u && threadIdx.x == 0u) a.work_count[0] = 0u;     // the repair list of this launch (stitch appends, repair reads): zeroed here, two launches ahead
    const uint32_t lane = threadIdx.x;
    const uint32_t g = blockIdx.x % a.n_groups, d = blockIdx.x / a.n_groups;
    const uint32_t b = g * 64u + lane, bc = min(b, a.B - 1u);
    const bool active = b < a.B;
    const uint32_t Bpad = a.n_groups * 64u, rows = a.tiles_per_chunk * 64u, C = a.C;
    const uint32_t S = a.n_states[d];
    const uint32_t n = min(a.lens[bc], a.M);
    const uint32_t dead = S + 1u;             // table rows: real states 0 .. S - 1, the dummy row S, the dead row S + 1
    constexpr bool small = SMALL;
    const size_t row_step = (size_t)a.D * Bpad * a.row_bytes;                      // chunk k + 1's row of the same (def, string)
    const uint8_t *row0 = a.rows + ((size_t)d * Bpad + bc) * a.row_bytes;          // chunk 0's
    struct Row { uint4 c0, c1, q0, q1; };
    auto load_row = [&](const uint32_t k) -> Row {
        const uint8_t *rp = row0 + (size_t)min(k, C - 1u) * row_step;
        Row r;
        r.c0 = make_uint4(0, 0, 0, 0); r.c1 = r.c0;
        if (small) {
            r.c0 = *reinterpret_cast<const uint4 *>(rp);
            r.c1 = *reinterpret_cast<const uint4 *>(rp + (a.smax > 16u ? 16 : 0));      // (unconditional: a load, not a branch)
        }
        r.q0 = *reinterpret_cast<const uint4 *>(rp + a.smax);
        r.q1 = *reinterpret_cast<const uint4 *>(rp + a.smax + 16);
        return r;
    };
    Row ring[kAheadRows];
#pragma unroll
    for (uint32_t u = 0; u < kAheadRows; ++u) ring[u] = load_row(u);
    // the init words trail the chain: chunk k's word is written when its slot comes round again (its tag load was issued kAheadRows steps ago)
    uint32_t pend_state[kAheadRows], pend_tag[kAheadRows];
    auto write_init = [&](const uint32_t k, const uint32_t st, const uint32_t tag) {
        if (active && !(a.dbg & 2u)) a.init[((size_t)k * a.B + b) * a.D + d] = st | (tag & 0xffu) << 16 | ((tag >> 9) & 1u) << 24;
    };
    uint32_t s = a.dc[d].first_state, prev = 0xffffffffu;
    for (uint32_t k0 = 0; k0 < C; k0 += kAheadRows) {
#pragma unroll
        for (uint32_t u = 0; u < kAheadRows; ++u) {
            const uint32_t k = k0 + u;
            if (k < C) {
                if (k >= kAheadRows) write_init(k - kAheadRows, pend_state[u], pend_tag[u]);
                const uint32_t r0 = k * rows;
                // ---- chunk k's init word: the state at its first row (rows beyond n hold the dummy state, table row S: lib.rs:413; row n itself
                // holds s[n]) and the substr id / end flag of the transition into that row
                const bool tagged = prev < S && s < S && r0 <= n && !(a.dbg & 1u);
                const uint32_t tag = a.pair_tags[d][tagged ? (size_t)prev * S + s : (size_t)0];     // (unconditional: a load, not a branch)
                pend_state[u] = r0 > n ? S : s; pend_tag[u] = tagged ? tag : 0u;
                // ---- through chunk k
                const Row rw = ring[u];
                ring[u] = load_row(k + kAheadRows);
                const bool ends_here = r0 + rows > n;          // the string ends in this chunk: every later chunk is padding (its start state is never looked at)
                const bool is_dead = s >= S;                   // an undefined transition further up (lib.rs:817): absorbing
                uint32_t ns = dead, np = dead;
                {
                    const uint32_t cw[8] = {rw.c0.x, rw.c0.y, rw.c0.z, rw.c0.w, rw.c1.x, rw.c1.y, rw.c1.z, rw.c1.w};
                    const uint32_t w[6] = {rw.q0.x, rw.q0.y, rw.q0.z, rw.q0.w, rw.q1.x, rw.q1.y};
                    uint32_t v;                                // the state after stage A in this chunk: one of its keys
                    if (small) {
                        uint32_t word = cw[0];
#pragma unroll
                        for (uint32_t j = 1; j < 8u; ++j) if ((s >> 2) == j) word = cw[j];
                        v = (word >> (8u * (s & 3u))) & 0xffu;
                    } else {
                        v = (row0 + (size_t)k * row_step)[min(s, S - 1u)];
                    }
#pragma unroll
                    for (uint32_t j = 0; j < kSpecKeys; ++j) {
                        const uint32_t kj = (w[j >> 2] >> (8u * (j & 3u))) & 0xffu;
                        if (kj == v) { ns = (w[2u + (j >> 2)] >> (8u * (j & 3u))) & 0xffu; np = (w[4u + (j >> 2)] >> (8u * (j & 3u))) & 0xffu; }
                    }
                }
                // the chunk's start states did not merge into the scout's bounds: the wave walks it for this lane
                uint64_t flagged = __ballot(active && !ends_here && !is_dead && rw.q1.z != 0u);
                while (flagged) {
                    const int j = __ffsll((unsigned long long)flagged) - 1;
                    flagged &= flagged - 1;
                    uint32_t sj = (uint32_t)__builtin_amdgcn_readlane((int)s, j), pj = 0;
                    spec_walk_flagged(a, g * 64u + (uint32_t)j, d, k, sj, pj, lane, smem);
                    if (lane == (uint32_t)j) { ns = sj; np = pj; }
                }
                if (ends_here) prev = 0xffffffffu;
                else if (is_dead) prev = s;
                else { s = ns >= S ? dead : ns; prev = np; }
            }
        }
    }
    // (slot u holds chunk k with k % kAheadRows == u: the last one of each slot)
#pragma unroll
    for (uint32_t u = 0; u < kAheadRows; ++u) {
        if (u < C) {
            const uint32_t last = ((C - 1u - u) / kAheadRows) * kAheadRows + u;
            write_init(last, pend_state[u], pend_tag[u]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// stitch: thread = string
// ---------------------------------------------------------------------------------------------
// Work item of the repair launch: chunk `k` of string `b` with the TRUE start_mask before its first row and the true end_mask of its
// last row.
struct SpecWork { uint32_t b, k_sm_em; };

// repair: one WAVE per work item, lanes = rows.  The masked rows of rows [r0, r1) of string b are recomputed from its finished
// records and the input bytes — App. A.3 (lib.rs:598-764) with the two "last event wins" scans as wave ballots + carry chains
// (hrx_lane.h fill_up / fill_down), 64 rows per step.
__global__ __launch_bounds__(64) void spec_repair_kernel(const SpecArgs a) {
    __shared__ uint64_t smbits[kSpecMaxChunkTiles];
    __shared__ uint32_t packed[kSpecMaxChunkTiles * 64u];      // per row of the chunk: sid | st << 10 | en1 << 11 | byte << 16
    const uint32_t lane = threadIdx.x;
    const uint32_t count = min(a.work_count[0], a.work_cap);
    const SpecWork *work = reinterpret_cast<const SpecWork *>(a.work);
    const uint32_t rows = a.tiles_per_chunk * 64u, D = a.D;
    for (uint32_t wi = blockIdx.x; wi < count; wi += gridDim.x) {
        const uint32_t b = work[wi].b, k = work[wi].k_sm_em & 0xffu;
        uint32_t sm = (work[wi].k_sm_em >> 8) & 1u;
        const uint32_t em_last = (work[wi].k_sm_em >> 9) & 1u;
        const uint32_t r0 = k * rows, r1 = min(r0 + rows, a.M), steps = (r1 - r0) / 64u;
        const uint32_t n = min(a.lens[b], a.M);
        const uint32_t blk0 = (b / kPmBlock) * kPmBlock, nb = min(kPmBlock, a.B - blk0), bl = b - blk0;
        const size_t q4 = (a.M + 3u) / 4u, q8 = (a.M + 7u) / 8u;
        const uint32_t *rec = a.records + (size_t)blk0 * q4 * D * 4u;
        uint16_t *msk = a.masked + (size_t)blk0 * q8 * 8u;
        const uint8_t *cptr = a.in_pm ? a.chars + (size_t)blk0 * a.stride + (size_t)bl * 16u : a.chars + (size_t)b * a.stride;
        const size_t cmul = a.in_pm ? (size_t)nb * 16u : (size_t)16u;
        // row r: sid = sum of the defs' substr ids, st = any start_enable, en1 = any end_enable (= EN[r + 1], lib.rs:501-519); zeros beyond M
        auto row = [&](const uint32_t r, uint32_t &sid, uint32_t &st, uint32_t &en1) {
            sid = 0; st = 0; en1 = 0;
            if (r >= a.M) return;
            for (uint32_t d = 0; d < D; ++d) {
                const uint32_t w = a.rec_planes[0] ? a.rec_planes[d][(size_t)blk0 * q4 * 4u + ((size_t)(r >> 2) * nb + bl) * 4u + (r & 3u)]
                                                   : rec[(((size_t)(r >> 2) * D + d) * nb + bl) * 4u + (r & 3u)];
                sid += (w >> 16) & 0xffu; st |= (w >> 24) & 1u; en1 |= (w >> 25) & 1u;
            }
        };
        // ---- the chunk's rows into LDS, eight steps' loads in flight at once (a pass that waited for every step's loads took 16 x 2
        // round trips per item: 41 us at D = 3)
        for (uint32_t s0 = 0; s0 < steps; s0 += 8u) {
            uint32_t pk[8];
#pragma unroll
            for (uint32_t u = 0; u < 8u; ++u) {
                const uint32_t p = r0 + min(s0 + u, steps - 1u) * 64u + lane;
                uint32_t sid, st, en1;
                row(p, sid, st, en1);
                const uint32_t c = cptr[(size_t)(p >> 4) * cmul + (p & 15u)];
                pk[u] = sid | st << 10 | en1 << 11 | c << 16;
            }
#pragma unroll
            for (uint32_t u = 0; u < 8u; ++u) if (s0 + u < steps) packed[(s0 + u) * 64u + lane] = pk[u];
        }
        // ---- forward: start_mask (lib.rs:598-645)
        uint32_t c_sid = 0, c_en = 0, t_st;
        if (r0 > 0u) row(r0 - 1u, c_sid, t_st, c_en);       // the row before the chunk (every lane loads the same word)
        uint32_t n_sid = 0, n_st = 0, t_en;
        row(r1, n_sid, n_st, t_en);                          // row r1 (zeros if r1 == M: SID[M] = 0, ST[M] = 0)
        __syncthreads();
        for (uint32_t s = 0; s < steps; ++s) {
            const uint32_t w = packed[s * 64u + lane];
            const uint32_t sid = w & 0x3ffu, st = (w >> 10) & 1u, en1 = (w >> 11) & 1u;
            uint32_t sidp = (uint32_t)__shfl_up((int)sid, 1, 64), en = (uint32_t)__shfl_up((int)en1, 1, 64);
            if (lane == 0u) { sidp = c_sid; en = c_en; }
            const bool chg = sid != sidp;
            const uint64_t setm = __ballot(st && chg), rstm = __ballot(!st && en && chg);
            const uint64_t bits = fill_up(setm, rstm, sm);
            sm = (uint32_t)(bits >> 63) & 1u;
            if (lane == 0u) smbits[s] = bits;
            c_sid = (uint32_t)__shfl((int)sid, 63, 64);
            c_en = (uint32_t)__shfl((int)en1, 63, 64);
        }
        __syncthreads();
        // ---- backward: end_mask (lib.rs:663-714); the event of position p is made of row p + 1's quantities
        uint32_t em = 0;                                     // e_M = 0; for r1 < M the chunk's last row is forced to em_last below
        for (uint32_t s = steps; s-- > 0u;) {
            const uint32_t p = r0 + s * 64u + lane;
            const uint32_t w = packed[s * 64u + lane];
            const uint32_t sid = w & 0x3ffu, st = (w >> 10) & 1u, enp1 = (w >> 11) & 1u, c = w >> 16;
            uint32_t sid1 = (uint32_t)__shfl_down((int)sid, 1, 64), st1 = (uint32_t)__shfl_down((int)st, 1, 64);
            if (lane == 63u) { sid1 = n_sid; st1 = n_st; }
            const bool chg = sid1 != sid;
            bool set = enp1 && chg, rst = !enp1 && st1 && chg;
            if (r1 < a.M && p + 1u == r1) { set = em_last != 0u; rst = em_last == 0u; }    // the stitch launch knows this row's end_mask
            const uint64_t bits = fill_down(__ballot(set), __ballot(rst), em);
            em = (uint32_t)bits & 1u;
            const uint32_t mask = (uint32_t)((smbits[s] >> lane) & (bits >> lane) & 1ull) && p < n ? 1u : 0u;
            msk[((size_t)(p >> 3) * nb + bl) * 8u + (p & 7u)] = mask ? (uint16_t)(c | sid << 8) : (uint16_t)0;
            n_sid = (uint32_t)__shfl((int)sid, 0, 64);
            n_st = (uint32_t)__shfl((int)st, 0, 64);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void spec_stitch_kernel(const SpecArgs a) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    const uint32_t rows = a.tiles_per_chunk * 64u, C = a.C;
    const uint32_t n = min(a.lens[b], a.M);
    // every chunk's two summary words, all loads in flight at once (a thread's chunks are 8 B x B apart: a loop that waited for each
    // word took 32 us for 32 chunks)
    uint2 vi[kSpecMaxChunks];
    uint64_t vst[kSpecMaxChunks];
#pragma unroll
    for (uint32_t k = 0; k < kSpecMaxChunks; ++k) {
        vi[k] = make_uint2(0, 0); vst[k] = 0;
        if (k < C) { vi[k] = a.vinfo[(size_t)k * a.B + b]; vst[k] = a.vstatus[(size_t)k * a.B + b]; }
    }
    // ---- status: lowest chunk's undefined transition (lib.rs:806-817), else the lowest overlap row, else ok with the accept bits of
    // the chunk that holds row n (n == M: the last chunk); a bad length shows in every chunk
    uint64_t sw_err = 0, sw_ov = 0, sw_acc = 0;
    bool have_err = false, have_ov = false;
#pragma unroll
    for (uint32_t k = 0; k < kSpecMaxChunks; ++k) {
        if (k < C) {
            const uint64_t w = vst[k];
            const uint32_t code = (uint32_t)(w & 0xffu);
            const uint32_t r0 = k * rows;
            // the reference walks the defs one after the other (lib.rs:806): the lowest DEF with an undefined transition anywhere wins,
            // at its first position — the first chunk that reports that def (a chunk reports its own lowest def)
            if (code == kStatusBadLength && !have_err) { sw_err = w; have_err = true; }
            if (code == kStatusInvalidTransition && (!have_err || ((sw_err & 0xffu) == kStatusInvalidTransition && ((w >> 8) & 0xffu) < ((sw_err >> 8) & 0xffu)))) { sw_err = w; have_err = true; }
            if (code == kStatusFlagOverlap && (!have_ov || (w >> 40) < (sw_ov >> 40))) { sw_ov = w; have_ov = true; }
            if (code == kStatusOk && ((n >= r0 && n < r0 + rows) || (k + 1u == C && n >= r0))) sw_acc = w;
        }
    }
    a.status[b] = have_err ? sw_err : have_ov ? sw_ov : sw_acc;
    if (have_err) return;                     // rows of a string whose code is not 0 are unspecified
    // ---- reveal-mask carries across the chunk borders
    // E[k] = end_mask of chunk k's last row = what the first deciding tile of a later chunk says
    uint32_t E_bits = 0, Enext = 0;
#pragma unroll
    for (uint32_t kk = 0; kk < kSpecMaxChunks; ++kk) {
        const uint32_t k = kSpecMaxChunks - 1u - kk;
        if (k < C) {
            uint32_t E = 0;
            if (k + 1u < C) {
                const uint32_t dec = (vi[(k + 1u) % kSpecMaxChunks].x >> 3) & 3u;
                E = dec == 1u ? 1u : dec == 2u ? 0u : Enext;
            }
            E_bits |= E << k;
            Enext = E;
        }
    }
    uint32_t sm = 0;
    bool live = true;
#pragma unroll
    for (uint32_t k = 0; k < kSpecMaxChunks; ++k) {
        if (k < C && live) {
            const uint32_t pend = vi[k].x & 1u, fwd = (vi[k].x >> 1) & 1u, sm_out = (vi[k].x >> 2) & 1u;
            const uint32_t r0 = k * rows;
            if (r0 >= n && r0 > 0u) { live = false; }        // padding chunks: no flags, masks all zero whatever the carries
            else {
                // the chunk assumed start_mask = 0 at its first row / wrote its trailing rows with end_mask = 1: a work item for the repair
                // launch (one wave each) with the true values
                if (sm || (pend && !((E_bits >> k) & 1u))) {
                    const uint32_t at = atomicAdd(a.work_count, 1u);
                    if (at < a.work_cap) a.work[at] = make_uint2(b, k | sm << 8 | ((E_bits >> k) & 1u) << 9);
                }
                sm = fwd ? sm_out : sm;
            }
        }
    }
}

hipError_t launch_spec_scout(const SpecArgs &a, int num_cus, hipStream_t stream) {
    static std::atomic<size_t> granted[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    const bool compact = a.cimage != nullptr;
    const uint32_t lds = compact ? a.cimage_bytes : a.table_bytes;
    hipError_t e = compact ? ensure_lds(spec_scout_kernel<true>, granted[(dev & 31) + 32], lds) : ensure_lds(spec_scout_kernel<false>, granted[dev & 31], lds);
    if (e != hipSuccess) return e;
    // 16 waves per CU (the lookups saturate the LDS pipe from there): workgroups of 4 waves while four copies of the table fit LDS, of 8 while two do, of 16 above
    const size_t waves = (size_t)a.n_groups * a.C;
    const size_t copies = std::max<size_t>(1, kLdsLimit / std::max<uint32_t>(lds, 1u));
    const size_t wpw = copies >= 4 ? 4 : copies >= 2 ? 8 : 16, per_cu = std::min<size_t>(copies, 16 / wpw);
    const size_t grid = std::min<size_t>((waves + wpw - 1) / wpw, (size_t)num_cus * per_cu);
    if (compact) hipLaunchKernelGGL(spec_scout_kernel<true>, dim3((unsigned)std::max<size_t>(grid, 1)), dim3((unsigned)(64 * wpw)), lds, stream, a);
    else hipLaunchKernelGGL(spec_scout_kernel<false>, dim3((unsigned)std::max<size_t>(grid, 1)), dim3((unsigned)(64 * wpw)), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_spec_compose(const SpecArgs &a, hipStream_t stream) {
    const size_t lds = 65u * (size_t)a.smax;        // the maps of a chunk the scout gave up on (spec_walk_flagged)
    static std::atomic<size_t> granted[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    const bool small = a.smax <= 32u;
    hipError_t e = small ? ensure_lds(spec_compose_kernel<true>, granted[dev & 63], lds) : ensure_lds(spec_compose_kernel<false>, granted[(dev & 31) + 32], lds);
    if (e != hipSuccess) return e;
    // one wave per (64 strings, def)
    if (small) hipLaunchKernelGGL(spec_compose_kernel<true>, dim3(a.n_groups * a.D), dim3(64), lds, stream, a);
    else hipLaunchKernelGGL(spec_compose_kernel<false>, dim3(a.n_groups * a.D), dim3(64), lds, stream, a);
    return hipGetLastError();
}

// (a.work_count[0] must be zero when the stitch launch starts: a memset node in front of it)
hipError_t launch_spec_stitch(const SpecArgs &a, int num_cus, hipStream_t stream) {
    hipLaunchKernelGGL(spec_stitch_kernel, dim3((a.B + 63u) / 64u), dim3(64), 0, stream, a);
    hipLaunchKernelGGL(spec_repair_kernel, dim3((unsigned)num_cus * 8u), dim3(64), 0, stream, a);
    return hipGetLastError();
}

}  // namespace hrx
