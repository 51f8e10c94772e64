// hrx_kernel_tp.hip — position-major -> string-major: the rows of a finished position-major batch written out in the layout a
// per-circuit fill loop indexes (one circuit's rows contiguous: records [B][pitch][D], masked [B][pitch]; src/lib.rs:387-519 fills
// one circuit at a time).
//
// String-major callers whose config has no fast string-major kernel of its own — more than three RegexDefs (walked in passes,
// hrx_kernel_mp.hip) and DFAs whose 4-byte table does not fit LDS (cfg 5's 256-state DFA: BYTE / HALF table kernels) — used to get
// per-lane 4-byte stores (3.7 ms for 65536 x 1024 rows at D = 5) or the one-wave global-table walk (0.21 of peak on cfg 5).  They
// now run the position-major path into context scratch and this kernel turns the rows around: pure streaming, 4 D + 2 bytes read
// and written per row, every global access a run of full lines.
//
// A workgroup takes a tile of 64 strings x R rows, R = 16 / 32 / 64 / 128 for D >= 10 / >= 3 / 2 / 1 (a string's R x D records are the run a store
// instruction's lanes write: at least 512 bytes; with 32 rows at D = 1 the 128-byte runs ran at 3.5 TB/s).  In: per (def, row quad) the 64 strings' 16-byte pieces are 1 KiB contiguous
// (lane = string), written to LDS as they come.  Out: a string's 32 rows x D records are 128 D contiguous bytes = D lines; a
// wave store writes one line of each of 8 strings (8 lanes x 16 B per string), and every lane gathers its four dwords — four
// (row, def) cells — from the LDS planes; lanes of different strings fall into different banks.
#include <hip/hip_runtime.h>

#include "hrx_device.h"

namespace hrx {

template <uint32_t kTpRows>
__global__ __launch_bounds__(256) void transpose_pm_to_sm_kernel(const TransposeArgs a) {
    const uint32_t D = a.D, M = a.M, B = a.B;
    const uint32_t tiles_r = (M + kTpRows - 1u) / kTpRows;
    const uint32_t tile = blockIdx.x;
    const uint32_t g = tile / tiles_r, tr = tile % tiles_r;      // string group of 64, row tile of 32
    const uint32_t b0 = g * 64u, row0 = tr * kTpRows;
    const uint32_t blk0 = (b0 / kPmBlock) * kPmBlock, nb = min(kPmBlock, B - blk0);
    const size_t q4 = (M + 3u) / 4u, q8 = (M + 7u) / 8u;
    const uint32_t nq = min(kTpRows / 4u, (uint32_t)q4 - row0 / 4u);          // row quads of this tile that exist
    const uint32_t tid = threadIdx.x, s = tid & 63u, part = tid >> 6;
    const uint32_t bl = min(b0 + s, B - 1u) - blk0;
    // ---- in: records [q4][D][nb][4] and masked [q8][nb][8] of the tile -> LDS planes [def][quad][string] (16 B each), masked [octet][string]
    const uint32_t msk_base = D * (kTpRows / 4u) * 64u * 16u;
    {
        const uint4 *rec = reinterpret_cast<const uint4 *>(a.records_pm) + (size_t)blk0 * q4 * D;
        for (uint32_t i = part; i < D * nq; i += 4u) {
            const uint32_t q = i / D, d = i % D;
            const uint4 v = rec[((size_t)(row0 / 4u + q) * D + d) * nb + bl];
            *reinterpret_cast<uint4 *>(smem + ((d * (kTpRows / 4u) + q) * 64u + s) * 16u) = v;
        }
        const uint4 *msk = reinterpret_cast<const uint4 *>(a.masked_pm) + (size_t)blk0 * q8;
        const uint32_t no = min(kTpRows / 8u, (uint32_t)q8 - row0 / 8u);
        for (uint32_t o = part; o < no; o += 4u)
            *reinterpret_cast<uint4 *>(smem + msk_base + (o * 64u + s) * 16u) = msk[(size_t)(row0 / 8u + o) * nb + bl];
    }
    __syncthreads();
    // ---- out: records.  Item = (8 strings, one 128-byte line of each string's 128 D tile bytes); lane -> string l & 7, 16-byte unit l >> 3
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t rows_here = min(kTpRows, M - row0);
    const uint32_t inv = (65536u + D - 1u) / D;                 // i / D = (i * inv) >> 16 for i < 1024 (D <= 32)
    const uint32_t lines = (kTpRows * D + 31u) / 32u;           // 128-byte lines of a string's [R][D] tile (the last one half a line at R = 16 and odd D)
    for (uint32_t item = wave; item < 8u * lines; item += 4u) {
        const uint32_t sg = item / lines, line = item % lines;
        const uint32_t sl = sg * 8u + (lane & 7u), w = lane >> 3;
        const uint32_t b = b0 + sl;
        const uint32_t i0 = line * 32u + w * 4u;                // first of this lane's four dwords inside the string's [32][D] tile
        uint32_t v[4];
#pragma unroll
        for (uint32_t t = 0; t < 4u; ++t) {
            const uint32_t i = min(i0 + t, kTpRows * D - 1u), r = (i * inv) >> 16, d = i - r * D;
            v[t] = *reinterpret_cast<const uint32_t *>(smem + ((d * (kTpRows / 4u) + (r >> 2)) * 64u + sl) * 16u + (r & 3u) * 4u);
        }
        if (b < B && i0 < rows_here * D) {
            uint32_t *out = a.records + ((size_t)b * a.rec_pitch + row0) * D + i0;
            if (i0 + 4u <= rows_here * D) *reinterpret_cast<uint4 *>(out) = make_uint4(v[0], v[1], v[2], v[3]);
            else
                for (uint32_t t = 0; t < 4u && i0 + t < rows_here * D; ++t) out[t] = v[t];
        }
    }
    // ---- out: masked rows, 2 R bytes per string and tile: R / 8 lanes x 16 B per string, 512 / R strings per wave store
    constexpr uint32_t kOct = kTpRows / 8u, kStr = 64u / kOct;
    for (uint32_t item = wave; item < kOct; item += 4u) {
        const uint32_t sl = item * kStr + lane / kOct, o = lane % kOct;       // octet o of the tile (a string's octets in adjacent lanes: one contiguous run)
        const uint32_t b = b0 + sl;
        const uint4 v = *reinterpret_cast<const uint4 *>(smem + msk_base + (o * 64u + sl) * 16u);
        if (b < B && o * 8u < rows_here) {
            uint16_t *out = a.masked + (size_t)b * a.msk_pitch + row0 + o * 8u;
            if (o * 8u + 8u <= rows_here) *reinterpret_cast<uint4 *>(out) = v;
            else {
                const uint16_t h[8] = {(uint16_t)v.x, (uint16_t)(v.x >> 16), (uint16_t)v.y, (uint16_t)(v.y >> 16), (uint16_t)v.z, (uint16_t)(v.z >> 16), (uint16_t)v.w, (uint16_t)(v.w >> 16)};
                for (uint32_t t = 0; t < 8u && o * 8u + t < rows_here; ++t) out[t] = h[t];
            }
        }
    }
}

template <uint32_t R>
static hipError_t launch_tp(const TransposeArgs &a, hipStream_t stream) {
    const size_t lds = (size_t)a.D * (R / 4u) * 64u * 16u + (R / 8u) * 64u * 16u;
    static std::atomic<size_t> granted[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipError_t e = ensure_lds(transpose_pm_to_sm_kernel<R>, granted[dev & 63], lds);
    if (e != hipSuccess) return e;
    const size_t tiles = (((size_t)a.B + 63) / 64) * (((size_t)a.M + R - 1) / R);
    hipLaunchKernelGGL(transpose_pm_to_sm_kernel<R>, dim3((unsigned)tiles), dim3(256), lds, stream, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// The INPUT the other way round: string-major bytes (one contiguous &[u8] per string: what the reference's caller holds, src/lib.rs:311-315) ->
// HRX_LAYOUT_INPUT_POSITION_MAJOR ([stride/16][nb][16] per block of kPmBlock strings).  A workgroup turns a tile of 64 strings x 16 units of 16 bytes
// around through 16 KiB of LDS: in, a wave load reads 256-byte runs of four strings (lane = string * 16 + unit); out, a wave store writes one unit of all
// 64 strings, 1 KiB contiguous (lane = string).  The LDS rows are 17 units long: the column reads of the way out fall into different banks.
// Pure streaming: stride bytes read and written per string.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chars_sm_to_pm_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const uint32_t B, const uint32_t units /* stride / 16 */) {
    const uint32_t tiles_u = (units + 15u) / 16u;
    const uint32_t g = blockIdx.x / tiles_u, tu = blockIdx.x % tiles_u;
    const uint32_t b0 = g * 64u, u0 = tu * 16u;
    const uint32_t blk0 = (b0 / kPmBlock) * kPmBlock, nb = min(kPmBlock, B - blk0);
    const uint32_t tid = threadIdx.x;
    const uint4 *src = reinterpret_cast<const uint4 *>(in);
    uint4 *dst = reinterpret_cast<uint4 *>(out);
    uint4 *tile = reinterpret_cast<uint4 *>(smem);                 // [64 strings][17 units]
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) {                             // 1024 units of the tile, 256 threads
        const uint32_t i = k * 256u + tid, sl = i >> 4, u = i & 15u;
        const uint32_t b = min(b0 + sl, B - 1u), uu = min(u0 + u, units - 1u);
        tile[sl * 17u + u] = src[(size_t)b * units + uu];
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) {
        const uint32_t i = k * 256u + tid, u = i >> 6, sl = i & 63u;
        const uint32_t b = b0 + sl;
        if (b < B && u0 + u < units) dst[(size_t)blk0 * units + (size_t)(u0 + u) * nb + (b - blk0)] = tile[sl * 17u + u];
    }
}

hipError_t launch_chars_to_position_major(const uint8_t *chars_sm, size_t stride, size_t B, uint8_t *chars_pm, hipStream_t stream) {
    if (B == 0) return hipSuccess;
    const uint32_t units = (uint32_t)(stride / 16);
    const size_t tiles = ((B + 63) / 64) * ((units + 15u) / 16u);
    hipLaunchKernelGGL(chars_sm_to_pm_kernel, dim3((unsigned)tiles), dim3(256), 64 * 17 * 16, stream, chars_sm, chars_pm, (uint32_t)B, units);
    return hipGetLastError();
}

hipError_t launch_transpose(const TransposeArgs &a, hipStream_t stream) {
    if (a.B == 0 || a.M == 0) return hipSuccess;
    // D >= 10: 16-row tiles — the planes of a 32-row tile (8 D + 4 KiB) pass the 160 KiB of LDS at D = 20, and from D = 10 on leave room for one workgroup per CU only
    return a.D == 1 ? launch_tp<128>(a, stream) : a.D == 2 ? launch_tp<64>(a, stream) : a.D < 10 ? launch_tp<32>(a, stream) : launch_tp<16>(a, stream);
}

}  // namespace hrx
