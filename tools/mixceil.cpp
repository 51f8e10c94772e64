// mixceil.cpp — on-box probe (round 2): the NO-COMPUTE ceiling of the bench line's traffic mix on this box.
// The bench workload (65536 strings x 1024 rows, D = 1) reads 64 MiB of input and writes 256 MiB of records + 128 MiB of
// masked rows per launch = 469.8 MB (7 B/row).  Every variant below moves exactly those bytes and does nothing else:
//   copy      : plain grid-stride dwordx4 copy of the same total (224 MiB read + 224 MiB written) — the guide's figure
//   ratio     : 1 : 6 read : write, all three streams sequential, 16 B per lane per instruction, many waves — the most
//               memory-friendly way to move this byte mix (layout-free ceiling)
//   slab K    : the kernel's own address streams (position-major slabs: record quad q of all strings = 1 MiB, masked
//               octet = 1 MiB, input chunk = 1 MiB), one lane per string, issued by K waves per string-group that split the
//               rows between them (K = 1: one wave per SIMD like the shipped kernel's walkers; K = 2, 4: more waves in flight)
//   pair      : loader + walker waves like the shipped kernel: 4 waves per CU only read (16 B per lane per load, into a
//               register sink), 4 only write (record quad every step, masked octet every second step)
// Output: one line per variant, us per launch and TB/s over the 469.8 MB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr size_t B = 65536, M = 1024;
constexpr size_t IN_BYTES = B * M, REC_BYTES = B * M * 4, MSK_BYTES = B * M * 2, TOTAL = IN_BYTES + REC_BYTES + MSK_BYTES;

__global__ __launch_bounds__(256) void copy_k(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// 1 : 4 : 2 — per 16 input bytes a lane writes 64 B of records and 32 B of masked rows, everything sequential
__global__ __launch_bounds__(256) void ratio_k(const uint4 *__restrict__ in, uint4 *__restrict__ rec, uint4 *__restrict__ msk, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = in[i];
        const size_t blk = i / 64, l = i % 64;  // a wave's 64 lanes write 1-KiB contiguous runs
#pragma unroll
        for (int k = 0; k < 4; ++k) rec[(blk * 4 + k) * 64 + l] = v;
#pragma unroll
        for (int k = 0; k < 2; ++k) msk[(blk * 2 + k) * 64 + l] = v;
    }
}

// the kernel's slabs: lane = string; rows [r0, r1) of the group, quads of 4 rows
template <int K>
__global__ __launch_bounds__(256) void slab_k(const uint4 *__restrict__ in, uint4 *__restrict__ rec, uint4 *__restrict__ msk, unsigned *sink) {
    const size_t gw = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64, lane = threadIdx.x & 63;
    const size_t g = gw / K, k = gw % K;          // string group, row slice
    const size_t b = g * 64 + lane;
    const size_t q0 = k * (M / 4 / K), q1 = q0 + M / 4 / K;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t q = q0; q < q1; ++q) {
        if ((q & 3) == 0) { const uint4 v = in[(q >> 2) * B + b]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
        rec[q * B + b] = make_uint4((unsigned)q, 1, 2, 3);
        if (q & 1) msk[(q >> 1) * B + b] = make_uint4(0, 0, 0, (unsigned)q);
    }
    if (acc.x == 0x12345678u) sink[0] = acc.y ^ acc.z ^ acc.w;
}

// loader / walker split: waves 0..3 of a 512-thread workgroup write, waves 4..7 read (one workgroup per CU, 4 groups each)
__global__ __launch_bounds__(512) void pair_k(const uint4 *__restrict__ in, uint4 *__restrict__ rec, uint4 *__restrict__ msk, unsigned *sink) {
    const size_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t g = (size_t)blockIdx.x * 4 + (wave & 3), b = g * 64 + lane;
    if (wave >= 4) {
        uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll 8
        for (size_t c = 0; c < M / 16; ++c) { const uint4 v = in[c * B + b]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
        if (acc.x == 0x12345678u) sink[0] = acc.y ^ acc.z ^ acc.w;
        return;
    }
    for (size_t q = 0; q < M / 4; ++q) {
        rec[q * B + b] = make_uint4((unsigned)q, 1, 2, 3);
        if (q & 1) msk[(q >> 1) * B + b] = make_uint4(0, 0, 0, (unsigned)q);
    }
}

// variants of pair_k.  MODE bit 0: one merged output array, per octet of rows three 1-MiB planes [rec rows 0-3][rec rows 4-7][masked]
// (one compact write window instead of two); bit 1: non-temporal stores; bit 2: the readers' loads all hit the same
// 64 KiB (input "free": what the writers alone sustain in this structure); bit 3: writers wait (workgroup barrier) until
// the readers of their workgroup have all their loads back (reads first, then writes); bit 4: no writers (reads only)
template <int MODE>
__global__ __launch_bounds__(512) void pairv_k(const uint4 *__restrict__ in, uint4 *__restrict__ rec, uint4 *__restrict__ msk, unsigned *sink) {
    const size_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t g = (size_t)blockIdx.x * 4 + (wave & 3), b = g * 64 + lane;
    typedef unsigned v4 __attribute__((ext_vector_type(4)));
    if (wave >= 4) {
        uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll 16
        for (size_t c = 0; c < M / 16; ++c) {
            const uint4 v = (MODE & 4) ? in[(c & 3) * 1024 + lane + (wave & 3) * 64] : in[c * B + b];
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
        if (acc.x == 0x12345678u) sink[0] = acc.y ^ acc.z ^ acc.w;
        if (MODE & 8) __syncthreads();
        return;
    }
    if (MODE & 8) __syncthreads();
    if (MODE & 16) return;
    auto st = [&](uint4 *p, const uint4 v) {
        if (MODE & 2) __builtin_nontemporal_store(v4{v.x, v.y, v.z, v.w}, reinterpret_cast<v4 *>(p));
        else *p = v;
    };
    for (size_t q = 0; q < M / 4; ++q) {
        if (MODE & 1) {
            st(rec + ((q >> 1) * 3 + (q & 1)) * B + b, make_uint4((unsigned)q, 1, 2, 3));
            if (q & 1) st(rec + ((q >> 1) * 3 + 2) * B + b, make_uint4(0, 0, 0, (unsigned)q));
        } else {
            st(rec + q * B + b, make_uint4((unsigned)q, 1, 2, 3));
            if (q & 1) st(msk + (q >> 1) * B + b, make_uint4(0, 0, 0, (unsigned)q));
        }
    }
}


// pair_k with non-temporal stores, a slab pitch of B + PAD 16-byte units (PAD = 0: slabs exactly 1 MiB apart) and PACE x s_sleep 1
// (64 cycles each) per quad in the writers: what the write stream sustains when the walkers are slower than the memory
template <int PAD, int PACE>
__global__ __launch_bounds__(512) void pairp_k(const uint4 *__restrict__ in, uint4 *__restrict__ rec, uint4 *__restrict__ msk, unsigned *sink) {
    const size_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t g = (size_t)blockIdx.x * 4 + (wave & 3), b = g * 64 + lane;
    constexpr size_t P = B + PAD;
    typedef unsigned v4 __attribute__((ext_vector_type(4)));
    if (wave >= 4) {
        uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll 16
        for (size_t c = 0; c < M / 16; ++c) { const uint4 v = in[c * B + b]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
        if (acc.x == 0x12345678u) sink[0] = acc.y ^ acc.z ^ acc.w;
        return;
    }
    for (size_t q = 0; q < M / 4; ++q) {
        __builtin_nontemporal_store(v4{(unsigned)q, 1, 2, 3}, reinterpret_cast<v4 *>(rec + q * P + b));
        if (q & 1) __builtin_nontemporal_store(v4{0, 0, 0, (unsigned)q}, reinterpret_cast<v4 *>(msk + (q >> 1) * P + b));
#pragma unroll
        for (int i = 0; i < PACE; ++i) __builtin_amdgcn_s_sleep(1);
    }
}

// pair_k with the shipped kernel's store policy: streaming stores, except that the record quads of every other 64-row tile are
// stored write-back (plan_nt_mix, hrx_kernel.hip: ~128 MiB of a launch's records stay in the cache hierarchy)
__global__ __launch_bounds__(512) void pairmix_k(const uint4 *__restrict__ in, uint4 *__restrict__ rec, uint4 *__restrict__ msk, unsigned *sink) {
    const size_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t g = (size_t)blockIdx.x * 4 + (wave & 3), b = g * 64 + lane;
    typedef unsigned v4 __attribute__((ext_vector_type(4)));
    if (wave >= 4) {
        uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll 16
        for (size_t c = 0; c < M / 16; ++c) { const uint4 v = in[c * B + b]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
        if (acc.x == 0x12345678u) sink[0] = acc.y ^ acc.z ^ acc.w;
        return;
    }
    for (size_t q = 0; q < M / 4; ++q) {
        if ((q >> 4) & 1) rec[q * B + b] = make_uint4((unsigned)q, 1, 2, 3);
        else __builtin_nontemporal_store(v4{(unsigned)q, 1, 2, 3}, reinterpret_cast<v4 *>(rec + q * B + b));
        if (q & 1) __builtin_nontemporal_store(v4{0, 0, 0, (unsigned)q}, reinterpret_cast<v4 *>(msk + (q >> 1) * B + b));
    }
}

static bool g_brief = false;
template <class F> static void timeit(const char *name, F &&launch, const char *key = nullptr) {
    if (g_brief && !key) return;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> us;
    for (int i = 0; i < 5; ++i) launch();
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 40; ++i) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        us.push_back(ms * 1e3f / 40);
    }
    std::sort(us.begin(), us.end());
    if (g_brief) { printf("MIXCEIL %s %.2f\n", key, us[2]); return; }
    printf("%-34s median %.1f us (min %.1f max %.1f)  %.2f TB/s  = %.3f of 8 TB/s\n", name, us[2], us[0], us[4], TOTAL / (us[2] * 1e-6) / 1e12,
           TOTAL / (us[2] * 1e-6) / 8e12);
}

// --sustained: `n` back-to-back launches, one event pair per `chunk` launches -> the time series (does the rate hold once the
// chip has been at full memory load for tens of milliseconds?)
template <class F> static void series(const char *name, F &&launch, int n = 1600, int chunk = 40) {
    std::vector<hipEvent_t> ev(n / chunk + 1);
    for (auto &evt : ev) CK(hipEventCreate(&evt));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(ev[0], 0));
    for (int i = 0; i < n / chunk; ++i) {
        for (int k = 0; k < chunk; ++k) launch();
        CK(hipEventRecord(ev[i + 1], 0));
    }
    CK(hipDeviceSynchronize());
    printf("%-28s us/launch per %d launches:", name, chunk);
    for (int i = 0; i < n / chunk; ++i) { float ms; CK(hipEventElapsedTime(&ms, ev[i], ev[i + 1])); printf(" %.1f", ms * 1e3f / chunk); }
    printf("\n");
    for (auto &evt : ev) CK(hipEventDestroy(evt));
}

int main(int argc, char **argv) {
    g_brief = argc > 1 && std::string(argv[1]) == "--brief";   // bench.py: three machine-readable lines
    const bool sustained = argc > 1 && std::string(argv[1]) == "--sustained";
    uint4 *in, *rec, *msk, *cpy; unsigned *sink;
    CK(hipMalloc(&in, IN_BYTES)); CK(hipMalloc(&rec, REC_BYTES + MSK_BYTES + (64u << 20))); msk = rec + REC_BYTES / 16; /* one allocation: the merged variant uses it as one array */ CK(hipMalloc(&sink, 4)); CK(hipMalloc(&cpy, TOTAL));
    CK(hipMemset(in, 1, IN_BYTES));
    if (!g_brief) printf("bench-line traffic mix without compute: %.1f MB per launch (64 MiB read, 384 MiB written)\n", TOTAL / 1e6);
    if (argc > 2 && std::string(argv[1]) == "--rotate") {
        // N sets of (input, records, masked) used in turn: nothing a launch touches is left in the Infinity Cache by the launches before
        const int nsets = atoi(argv[2]);
        std::vector<uint4 *> ins(nsets), recs(nsets);
        for (int k = 0; k < nsets; ++k) { CK(hipMalloc(&ins[k], IN_BYTES)); CK(hipMemset(ins[k], 1, IN_BYTES)); CK(hipMalloc(&recs[k], REC_BYTES + MSK_BYTES)); }
        int turn = 0;
        printf("%d buffer sets used in turn (%.1f GB):\n", nsets, nsets * (double)TOTAL / 1e9);
        for (int round = 0; round < 2; ++round) {
            timeit("pair, write-back stores", [&] { const int k = turn++ % nsets; hipLaunchKernelGGL(pair_k, dim3(256), dim3(512), 0, 0, ins[k], recs[k], recs[k] + REC_BYTES / 16, sink); });
            timeit("pair, non-temporal stores", [&] { const int k = turn++ % nsets; hipLaunchKernelGGL(pairv_k<2>, dim3(256), dim3(512), 0, 0, ins[k], recs[k], recs[k] + REC_BYTES / 16, sink); });
            timeit("pair, streaming + every other tile's records write-back", [&] { const int k = turn++ % nsets; hipLaunchKernelGGL(pairmix_k, dim3(256), dim3(512), 0, 0, ins[k], recs[k], recs[k] + REC_BYTES / 16, sink); });
            timeit("copy 8192x256 (first set only)", [&] { hipLaunchKernelGGL(copy_k, dim3(8192), dim3(256), 0, 0, (const uint4 *)cpy, cpy + TOTAL / 32, TOTAL / 32); });
        }
        return 0;
    }
    if (sustained) {
        for (int round = 0; round < 2; ++round) {
            series("pair, non-temporal stores", [&] { hipLaunchKernelGGL(pairv_k<2>, dim3(256), dim3(512), 0, 0, in, rec, msk, sink); });
            series("copy 8192x256", [&] { hipLaunchKernelGGL(copy_k, dim3(8192), dim3(256), 0, 0, (const uint4 *)cpy, cpy + TOTAL / 32, TOTAL / 32); });
            series("pair, write-back stores", [&] { hipLaunchKernelGGL(pair_k, dim3(256), dim3(512), 0, 0, in, rec, msk, sink); });
        }
        return 0;
    }
    for (int round = 0; round < (g_brief ? 1 : 2); ++round) {
        timeit("copy 224 MiB -> 224 MiB, 2048x256", [&] { hipLaunchKernelGGL(copy_k, dim3(2048), dim3(256), 0, 0, (const uint4 *)cpy, cpy + TOTAL / 32, TOTAL / 32); });
        timeit("copy 224 MiB -> 224 MiB, 8192x256", [&] { hipLaunchKernelGGL(copy_k, dim3(8192), dim3(256), 0, 0, (const uint4 *)cpy, cpy + TOTAL / 32, TOTAL / 32); }, "copy");
        timeit("ratio 1:4:2 sequential, 2048x256", [&] { hipLaunchKernelGGL(ratio_k, dim3(2048), dim3(256), 0, 0, in, rec, msk, IN_BYTES / 16); });
        timeit("ratio 1:4:2 sequential, 1024x256", [&] { hipLaunchKernelGGL(ratio_k, dim3(1024), dim3(256), 0, 0, in, rec, msk, IN_BYTES / 16); });
        timeit("slab, 1 wave per group (4/CU)", [&] { hipLaunchKernelGGL(slab_k<1>, dim3(256), dim3(256), 0, 0, in, rec, msk, sink); });
        timeit("slab, 2 waves per group (8/CU)", [&] { hipLaunchKernelGGL(slab_k<2>, dim3(512), dim3(256), 0, 0, in, rec, msk, sink); });
        timeit("slab, 4 waves per group (16/CU)", [&] { hipLaunchKernelGGL(slab_k<4>, dim3(1024), dim3(256), 0, 0, in, rec, msk, sink); });
        timeit("pair: 4 reader + 4 writer waves/CU", [&] { hipLaunchKernelGGL(pair_k, dim3(256), dim3(512), 0, 0, in, rec, msk, sink); }, "pair");
        timeit("pair, merged output array", [&] { hipLaunchKernelGGL(pairv_k<1>, dim3(256), dim3(512), 0, 0, in, rec, msk, sink); });
        timeit("pair, non-temporal stores", [&] { hipLaunchKernelGGL(pairv_k<2>, dim3(256), dim3(512), 0, 0, in, rec, msk, sink); }, "pair_nt");
        timeit("pair, streaming + every other tile's records write-back", [&] { hipLaunchKernelGGL(pairmix_k, dim3(256), dim3(512), 0, 0, in, rec, msk, sink); }, "pair_mix");
        timeit("pair, merged + non-temporal", [&] { hipLaunchKernelGGL(pairv_k<3>, dim3(256), dim3(512), 0, 0, in, rec, msk, sink); });
        timeit("pair, input from L2 (writes only)", [&] { hipLaunchKernelGGL(pairv_k<4>, dim3(256), dim3(512), 0, 0, in, rec, msk, sink); });
        timeit("pair, merged, input from L2", [&] { hipLaunchKernelGGL(pairv_k<5>, dim3(256), dim3(512), 0, 0, in, rec, msk, sink); });
        timeit("pair, reads first then writes", [&] { hipLaunchKernelGGL(pairv_k<8>, dim3(256), dim3(512), 0, 0, in, rec, msk, sink); });
        timeit("pair, merged, reads first", [&] { hipLaunchKernelGGL(pairv_k<9>, dim3(256), dim3(512), 0, 0, in, rec, msk, sink); });
        {
            uint4 *mskp = rec + (REC_BYTES + (32u << 20)) / 16;   // padded pitches need a little more room per array
#define PP(PAD, PACE) timeit("pair nt, pitch +" #PAD " x16 B, pace " #PACE, [&] { hipLaunchKernelGGL((pairp_k<PAD, PACE>), dim3(256), dim3(512), 0, 0, in, rec, mskp, sink); })
            PP(0, 0); PP(0, 1); PP(0, 2); PP(0, 3); PP(0, 4); PP(0, 6);
            PP(16, 0); PP(64, 0); PP(256, 0); PP(272, 0); PP(1024, 0); PP(4096, 0); PP(64, 2); PP(256, 2); PP(272, 2); PP(4096, 2);
        }
        timeit("pair, reads only (no writers)", [&] { hipLaunchKernelGGL(pairv_k<16>, dim3(256), dim3(512), 0, 0, in, rec, msk, sink); });
    }
    if (g_brief) {   // the same probes over 8 buffer sets used in turn: what HBM alone sustains for this mix (nothing left in the Infinity Cache)
        const int nsets = 8;
        std::vector<uint4 *> ins(nsets), recs(nsets);
        for (int k = 0; k < nsets; ++k) { CK(hipMalloc(&ins[k], IN_BYTES)); CK(hipMemset(ins[k], 1, IN_BYTES)); CK(hipMalloc(&recs[k], REC_BYTES + MSK_BYTES)); }
        int turn = 0;
        timeit("", [&] { const int k = turn++ % nsets; hipLaunchKernelGGL(pairv_k<2>, dim3(256), dim3(512), 0, 0, ins[k], recs[k], recs[k] + REC_BYTES / 16, sink); }, "pair_nt_fresh");
        timeit("", [&] { const int k = turn++ % nsets; hipLaunchKernelGGL(pairmix_k, dim3(256), dim3(512), 0, 0, ins[k], recs[k], recs[k] + REC_BYTES / 16, sink); }, "pair_mix_fresh");
    }
    return 0;
}
