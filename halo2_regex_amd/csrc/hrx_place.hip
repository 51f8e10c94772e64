// hrx_place.hip — the measuring half of the placement-aware output allocator (hrx_alloc_output_pair, include/hrx.h;
// DESIGN.md §6) and the no-compute traffic pass behind the roofline diagnostics (hrx_traffic_pass_device).
//
// On an MI355X two concurrent write streams run at 5.5-6.4 TB/s together when both lie in the same CLASS of the physical
// address space and at 7.0-7.4 TB/s when they lie in different ones (four classes, selected by address bits >= 2^33; the
// first 64 GiB of a contiguous allocation are one class: profiles/r02_probes/placement/region_map2d.txt).
// A witness launch writes two such streams, records and masked rows, part k of the one while part k of the other; buffers
// allocated one after the other — what any process does — come from one neighbourhood, i.e. one class.  User space cannot see
// physical addresses, but it can MEASURE: the probe kernel writes two regions the way a launch does (time-aligned parts, the
// streams' byte ratio, 1-KiB pieces per wave) and the allocator keeps the masked-row candidate whose neighbourhood measures
// fastest against the records' neighbourhood.
//
// The probe times ITSELF (s_memrealtime, the 100-MHz clock every CU shares: earliest wave start to latest wave end), so that a
// profiler that serialises and instruments dispatches (rocprofv3 --kernel-trace) cannot blur the sub-millisecond
// differences the search ranks candidates by — round 2's event-timed probe mis-ranked under a kernel trace.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "hrx_device.h"

namespace hrx {

constexpr uint32_t kProbeWaves = 1024;   // 256 workgroups of 4 waves
constexpr uint32_t kProbeParts = 16;

__global__ __launch_bounds__(256) void placement_probe_kernel(unsigned char *rec, size_t rec_part, unsigned char *msk, size_t msk_part,
                                                              uint32_t msk_waves, uint32_t steps, unsigned long long *clk) {
    const unsigned long long t0 = wall_clock64();
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const bool on_msk = wave < msk_waves;
    const uint32_t w = on_msk ? wave : wave - msk_waves, nw = on_msk ? msk_waves : kProbeWaves - msk_waves;
    unsigned char *base = (on_msk ? msk : rec) + ((size_t)w << 10) + lane * 16u;
    const size_t part = on_msk ? msk_part : rec_part, window = (size_t)nw << 10;
    const uint4 v = make_uint4(0, 0, 0, 0);
    for (uint32_t k = 0; k < kProbeParts; ++k)
        for (uint32_t s = 0; s < steps; ++s) store16_nt(base + k * part + s * window, v);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores have been accepted by the memory system
    if (lane == 0u) {
        atomicMin(clk, t0);
        atomicMax(clk + 1, wall_clock64());
    }
}

// Microseconds (device clock) for one time-aligned two-stream write over the two regions (their contents are overwritten):
// median of three passes after a warm-up pass; negative on a HIP error.  D: the records hold 4 * D bytes per row against the
// masked rows' 2.  `clk`: 16 bytes of device scratch.  *bytes_written: what one pass wrote.
double placement_probe_us(void *rec, size_t rec_bytes, void *msk, size_t msk_bytes, uint32_t D, hipStream_t st, unsigned long long *clk, size_t *bytes_written) {
    uint32_t msk_waves = (uint32_t)((kProbeWaves * 2u) / (4u * D + 2u) + 64u) / 128u * 128u;   // the streams' byte ratio, in units of 128 waves
    msk_waves = std::min(std::max(msk_waves, 128u), 512u);
    if (D == 0u) msk_waves = kProbeWaves / 2u;   // two EQUAL streams: one record plane against another (hrx_alloc_output_planes)
    const size_t rec_part = rec_bytes / kProbeParts / 4096 * 4096, msk_part = msk_bytes / kProbeParts / 4096 * 4096;
    const size_t steps = std::min<size_t>({(size_t)128, rec_part / ((size_t)(kProbeWaves - msk_waves) << 10), msk_part / ((size_t)msk_waves << 10)});
    if (steps == 0 || !clk) return -1.0;
    if (bytes_written) *bytes_written = (size_t)kProbeParts * steps * ((size_t)kProbeWaves << 10);
    double us[4];
    for (int r = 0; r < 4; ++r) {
        if (hipMemsetAsync(clk, 0xff, 8, st) != hipSuccess || hipMemsetAsync(clk + 1, 0, 8, st) != hipSuccess) return -1.0;
        hipLaunchKernelGGL(placement_probe_kernel, dim3(kProbeWaves / 4), dim3(256), 0, st, (unsigned char *)rec, rec_part, (unsigned char *)msk, msk_part,
                           msk_waves, (uint32_t)steps, clk);
        unsigned long long h[2] = {0, 0};
        if (hipMemcpyAsync(h, clk, 16, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1.0;
        us[r] = h[1] > h[0] ? (double)(h[1] - h[0]) * 0.01 : -1.0;   // 100 MHz
        if (us[r] < 0) return -1.0;
    }
    std::sort(us + 1, us + 4);   // (pass 0 is the warm-up)
    return us[2];
}

// ---------------------------------------------------------------------------------------------
// No-compute traffic pass: the memory traffic of ONE position-major witness launch of this shape — the input read in the
// loader's 16-byte-per-lane chunks, D record planes and the masked rows written in the walker's / finisher's 1-KiB runs at the
// launch's own addresses, with the launch's store policy (plan_nt_mix) — and no DFA work at all.  Like the kernel: one
// workgroup per CU, four reader and four writer waves, one string per lane, groups of 64 strings strided over the writers.
// What this takes is the box's ceiling for the launch's byte mix on these very buffers (bench.py: roofline.mix_ceiling).
// ---------------------------------------------------------------------------------------------
struct TrafficArgs {
    const unsigned char *chars;
    uint64_t stride;
    uint32_t B, M, D;
    unsigned char *records, *masked;
    uint32_t nt_mix;
    uint32_t *sink;
    uint32_t spread;
    unsigned char *planes[kMaxDefsPerLaunch];   // record planes in buffers of their own (WitnessArgs::rec_planes); planes[0] == NULL: the interleaved `records`
    uint32_t stripes;                           // 2: one def's plane in two row stripes (WitnessArgs::rec_stripes): quad q in planes[q % 2] at slot q / 2
};

// MODE 0: the interleaved records (the code of rounds 2-5, untouched: a run-time choice of the address form per store made the pass of the bench line 5 us slower than the launch it is the
// ceiling of); 1: record planes; 2: one def in two row stripes
template <int MODE>
__global__ __launch_bounds__(512) void traffic_pass_kernel(const TrafficArgs a) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t n_groups = (a.B + 63u) / 64u;
    const size_t q4 = (a.M + 3u) / 4u, q8 = (a.M + 7u) / 8u;
    // groups per workgroup: four adjacent ones (the loader / walker / finisher kernel's dealing) when the batch has at least four per CU; a smaller batch is SPREAD
    // over the CUs like the def-parallel kernel's two groups per workgroup (cfg 4's 512 groups ran on 128 of the 256 CUs here until round 5: its "ceiling" was slower than the kernel)
    const bool spread = a.spread != 0u;
    for (uint32_t g = spread ? blockIdx.x + (wave & 3u) * gridDim.x : blockIdx.x * 4u + (wave & 3u); g < n_groups; g += gridDim.x * 4u) {
        const uint32_t b = min(g * 64u + lane, a.B - 1u);
        const uint32_t blk0 = (g * 64u / kPmBlock) * kPmBlock, nb = min(kPmBlock, a.B - blk0), bl = b - blk0;
        if (wave >= 4u) {   // reader: every 16-byte chunk of the string
            const unsigned char *cp = a.chars + (size_t)blk0 * a.stride + (size_t)bl * 16u;
            uint4 acc = make_uint4(0, 0, 0, 0);
            const uint32_t nchunk = (uint32_t)(a.stride / 16u);
#pragma unroll 8
            for (uint32_t c = 0; c < nchunk; ++c) {
                const uint4 v = *reinterpret_cast<const uint4 *>(cp + (size_t)c * nb * 16u);
                acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
            }
            if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) a.sink[0] = acc.z ^ acc.w;
            continue;
        }
        unsigned char *rp = a.records + ((size_t)blk0 * q4 * a.D + bl) * 16u;
        unsigned char *mp = a.masked + ((size_t)blk0 * q8 + bl) * 16u;
        const uint32_t wb_k = a.nt_mix & 0xffu;
        // (round 5 also split a group's stores over TWO writer waves — four storing waves per CU for cfg 4's two groups: 0.69-0.70 against 0.70-0.71, no gain: it is the memory
        // system, not the number of storing waves — profiles/r05_probes/cfg4_front_width.txt)
        for (uint32_t q = 0; q < (uint32_t)q4; ++q) {
            const uint32_t t = q >> 4;
            const bool wb = wb_k != 0u && (t % wb_k) == wb_k - 1u;
            const uint4 v = make_uint4(q, 1, 2, 3);
            for (uint32_t d = 0; d < a.D; ++d) {
                unsigned char *p = MODE == 2 ? a.planes[q & 1u] + ((size_t)blk0 * ((q4 + 1u) / 2u) + bl + (size_t)(q >> 1) * nb) * 16u
                                 : MODE == 1 ? a.planes[d] + ((size_t)blk0 * q4 + bl + (size_t)q * nb) * 16u : rp + ((size_t)q * a.D + d) * nb * 16u;
                if (wb) *reinterpret_cast<uint4 *>(p) = v;
                else store16_nt(p, v);
            }
            if ((q & 1u) && (q >> 1) < (uint32_t)q8) store16_nt(mp + (size_t)(q >> 1) * nb * 16u, make_uint4(0, 0, 0, q));
        }
        if ((q4 & 1u) && (q4 >> 1) < q8) store16_nt(mp + (size_t)(q4 >> 1) * nb * 16u, make_uint4(0, 0, 0, 0));   // odd number of quads: the last octet
    }
}

// Record planes of a batch that leaves walker slots empty (at most two groups per CU: cfg 4) are written by the DEF-PARALLEL kernel: a wave per def stores its def's plane, the combiner wave
// the masked rows, a loader reads the input.  The pass for that launch deals its waves the same way — G groups per workgroup x (D plane writers + a masked-row writer + a reader) —
// because WHO issues the stores matters here: one wave storing all planes of a group in row order reads 0.75 of peak over planes on which the launch itself runs at 0.80
// (profiles/r06_probes/plane_select_cfg4.txt), and over interleaved records the per-plane dealing reads 0.68 where the one-wave pass reads 0.75 (plane_probe.txt).
__global__ __launch_bounds__(640) void traffic_pass_planes_pmd_kernel(const TrafficArgs a, const uint32_t G) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t W = a.D + 2u;                         // waves per group
    const uint32_t lg = wave / W, role = wave % W;        // role d < D: plane d; D: the masked rows; D + 1: the reader
    const uint32_t n_groups = (a.B + 63u) / 64u;
    const size_t q4 = (a.M + 3u) / 4u, q8 = (a.M + 7u) / 8u;
    for (uint32_t g = blockIdx.x * G + lg; g < n_groups; g += gridDim.x * G) {
        const uint32_t b = min(g * 64u + lane, a.B - 1u);
        const uint32_t blk0 = (g * 64u / kPmBlock) * kPmBlock, nb = min(kPmBlock, a.B - blk0), bl = b - blk0;
        if (role == a.D + 1u) {
            const unsigned char *cp = a.chars + (size_t)blk0 * a.stride + (size_t)bl * 16u;
            uint4 acc = make_uint4(0, 0, 0, 0);
            const uint32_t nchunk = (uint32_t)(a.stride / 16u);
#pragma unroll 8
            for (uint32_t c = 0; c < nchunk; ++c) {
                const uint4 v = *reinterpret_cast<const uint4 *>(cp + (size_t)c * nb * 16u);
                acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
            }
            if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) a.sink[0] = acc.z ^ acc.w;
        } else if (role == a.D) {
            unsigned char *mp = a.masked + ((size_t)blk0 * q8 + bl) * 16u;
            for (uint32_t o = 0; o < (uint32_t)q8; ++o) store16_nt(mp + (size_t)o * nb * 16u, make_uint4(0, 0, 0, o));
        } else {
            unsigned char *p = a.planes[role] + ((size_t)blk0 * q4 + bl) * 16u;
            for (uint32_t q = 0; q < (uint32_t)q4; ++q) store16_nt(p + (size_t)q * nb * 16u, make_uint4(q, 1, 2, 3));
        }
    }
}

// The same for STRING-MAJOR outputs, the way the walker/storer kernel (hrx_kernel_sm.hip) moves them: a store instruction writes the 128-byte lines of EIGHT strings
// (lane = string it * 8 + lane / 8, 16-byte piece lane % 8 of the line), 2 D lines of records and one line of masked rows per string and 64 rows; the input is read
// one string per lane, 16 bytes at a time, a stride apart.  rec_pitch / msk_pitch in rows.
struct TrafficSmArgs {
    const unsigned char *chars;
    uint64_t stride;
    uint32_t B, M, D, rec_pitch, msk_pitch;
    unsigned char *records, *masked;
    uint32_t nt_mix;
    uint32_t *sink;
};

__global__ __launch_bounds__(512) void traffic_pass_sm_kernel(const TrafficSmArgs a) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t n_groups = (a.B + 63u) / 64u;
    const uint32_t nblk = (a.M + 63u) / 64u;
    for (uint32_t g = blockIdx.x * 4u + (wave & 3u); g < n_groups; g += gridDim.x * 4u) {
        const uint32_t b0 = g * 64u;
        if (wave >= 4u) {   // reader: every 16-byte chunk of the lane's string
            const unsigned char *cp = a.chars + (size_t)min(b0 + lane, a.B - 1u) * a.stride;
            uint4 acc = make_uint4(0, 0, 0, 0);
            const uint32_t nchunk = (uint32_t)(a.stride / 16u);
#pragma unroll 8
            for (uint32_t c = 0; c < nchunk; ++c) {
                const uint4 v = *reinterpret_cast<const uint4 *>(cp + (size_t)c * 16u);
                acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
            }
            if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) a.sink[0] = acc.z ^ acc.w;
            continue;
        }
        const uint32_t js0 = lane >> 3, w = lane & 7u;
        const uint32_t wb_k = a.nt_mix & 0xffu;
        const size_t rec_row = (size_t)a.rec_pitch * a.D * 4u, msk_row = (size_t)a.msk_pitch * 2u;
        for (uint32_t t = 0; t < nblk; ++t) {
            const uint32_t rows = min(64u, a.M - t * 64u);
            const uint32_t rec_bytes = rows * a.D * 4u, msk_bytes = rows * 2u;      // of this block, per string
            const bool wb = wb_k != 0u && (t % wb_k) == wb_k - 1u;
            const uint4 v = make_uint4(t, 1, 2, 3);
            for (uint32_t j = 0; j * 128u < rec_bytes; ++j) {
                const uint32_t off = j * 128u + w * 16u;
#pragma unroll
                for (uint32_t it = 0; it < 8u; ++it) {
                    const uint32_t b = b0 + it * 8u + js0;
                    if (b < a.B && off < rec_bytes) {
                        unsigned char *p = a.records + (size_t)b * rec_row + (size_t)t * 64u * a.D * 4u + off;
                        if (wb) *reinterpret_cast<uint4 *>(p) = v; else store16_nt(p, v);
                    }
                }
            }
#pragma unroll
            for (uint32_t it = 0; it < 8u; ++it) {
                const uint32_t b = b0 + it * 8u + js0;
                if (b < a.B && w * 16u < msk_bytes) store16_nt(a.masked + (size_t)b * msk_row + (size_t)t * 128u + w * 16u, v);
            }
        }
    }
}

hipError_t launch_traffic_pass_sm(const uint8_t *chars, size_t stride, size_t B, size_t M, uint32_t D, uint32_t *records, size_t rec_pitch, uint16_t *masked,
                                  size_t msk_pitch, uint32_t nt_mix, uint32_t *sink, int num_cus, hipStream_t stream) {
    TrafficSmArgs a{chars, stride, (uint32_t)B, (uint32_t)M, D, (uint32_t)rec_pitch, (uint32_t)msk_pitch, (unsigned char *)records, (unsigned char *)masked, nt_mix, sink};
    const size_t n_groups = (B + 63) / 64, need = (n_groups + 3) / 4;
    const int grid = (int)std::min<size_t>(need, (size_t)num_cus);
    hipLaunchKernelGGL(traffic_pass_sm_kernel, dim3(grid < 1 ? 1 : grid), dim3(512), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_traffic_pass(const uint8_t *chars, size_t stride, size_t B, size_t M, uint32_t D, uint32_t *records, uint16_t *masked,
                               uint32_t nt_mix, uint32_t *sink, int num_cus, hipStream_t stream, uint32_t *const *planes, uint32_t stripes) {
    TrafficArgs a{chars, stride, (uint32_t)B, (uint32_t)M, D, (unsigned char *)records, (unsigned char *)masked, nt_mix, sink, 0u, {}, planes && D == 1 ? stripes : 1u};
    if (planes)
        for (uint32_t d = 0; d < D * a.stripes && d < kMaxDefsPerLaunch; ++d) a.planes[d] = (unsigned char *)planes[d];
    const size_t n_groups = (B + 63) / 64;
    a.spread = n_groups < (size_t)num_cus * 4 ? 1u : 0u;
    const size_t need = a.spread ? n_groups : (n_groups + 3) / 4;
    const int grid = (int)std::min<size_t>(need, (size_t)num_cus);
    const dim3 g(grid < 1 ? 1 : grid), b(512);
    // (what launches this shape with record planes: the def-parallel kernel — two and three defs at up to two groups per CU, four to eight defs always; hrx_kernel.hip)
    const bool pmd = a.planes[0] && a.stripes != 2u && D >= 2u && ((D <= 3u && n_groups <= (size_t)num_cus * 2 && B <= kPmBlock) || D >= 4u);
    if (pmd) {
        const uint32_t G = (D <= 3u && n_groups > (size_t)num_cus) ? 2u : 1u;
        const size_t needg = (n_groups + G - 1) / G;
        hipLaunchKernelGGL(traffic_pass_planes_pmd_kernel, dim3((unsigned)std::min<size_t>(std::max<size_t>(needg, 1), (size_t)num_cus)), dim3(64u * G * (D + 2u)), 0, stream, a, G);
        return hipGetLastError();
    }
    if (a.stripes == 2u) hipLaunchKernelGGL(traffic_pass_kernel<2>, g, b, 0, stream, a);
    else if (a.planes[0]) hipLaunchKernelGGL(traffic_pass_kernel<1>, g, b, 0, stream, a);
    else hipLaunchKernelGGL(traffic_pass_kernel<0>, g, b, 0, stream, a);
    return hipGetLastError();
}

}  // namespace hrx
