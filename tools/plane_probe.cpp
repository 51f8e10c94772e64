// plane_probe.cpp — on-box probe (round 6): does cfg 4's no-compute pass (0.70-0.71 of peak where cfg 2's reaches 0.77) follow WHERE its three
// record planes lie?  A D = 3 launch writes 12 of its 14 output bytes per row into ONE records allocation ([M/4][D][B][4]: the planes of a quad row
// 512 KiB apart) and 2 into the masked rows; the placement walk (csrc/hrx_place.hip) can only choose the masked rows' neighbourhood.  NOTES §4.3:
// two write streams in one class of the physical address space run at 5.5-6.4 TB/s together, in different classes at 6.6-7.4.
//
// What it does (no DFA work anywhere; every store is the walkers' 1-KiB run per wave, streaming):
//   1. allocates the input, one interleaved records buffer R, `cands` single-plane candidates P[i] ([M/4][B][4] each) and masked-row candidates;
//   2. pair matrix: two equal write streams over P[i] and P[j] (time-aligned parts, like placement_probe_kernel) -> GB/s, the class structure;
//   3. the launch's traffic (readers + writers dealt like the def-parallel kernel: 2 groups per CU) over
//        a. R + masked candidate k (today's layout), for every k;
//        b. random draws of three DIFFERENT plane candidates + a masked candidate (planes as separate allocations), many draws;
//        c. the draws with the lowest / highest pair rates of (2.) (all four streams in one class / spread over the classes);
//   4. the same interleaved pass with 1 / 2 / 4 storing waves per group (2 / 4 / 8 per CU) at M = rows and M = rows / 4.
// Output: text lines, ms per pass and the fraction of 8 TB/s over the algorithmic bytes B x M x (1 + 4 D + 2).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static __device__ __forceinline__ void store16_nt(void *p, const uint4 &v) {
    typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(p), "v"(v4u32{v.x, v.y, v.z, v.w}) : "memory");
}

// ---- two equal write streams, 512 waves each, 16 time-aligned parts (csrc/hrx_place.hip placement_probe_kernel) ----
__global__ __launch_bounds__(256) void pair_k(unsigned char *a, unsigned char *b, size_t part, uint32_t steps, unsigned long long *clk) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const bool on_b = wave & 1u;
    const uint32_t w = wave >> 1;
    unsigned char *base = (on_b ? b : a) + ((size_t)w << 10) + lane * 16u;
    const size_t window = (size_t)512 << 10;
    const uint4 v = make_uint4(0, 0, 0, 0);
    for (uint32_t k = 0; k < 16u; ++k)
        for (uint32_t s = 0; s < steps; ++s) store16_nt(base + k * part + s * window, v);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0u) {
        atomicMin(clk, t0);
        atomicMax(clk + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    }
}

static double pair_gbs(void *a, void *b, size_t bytes, unsigned long long *clk, hipStream_t st) {
    const size_t part = bytes / 16 / 4096 * 4096;
    const uint32_t steps = (uint32_t)std::min<size_t>(64, part / ((size_t)512 << 10));
    double us[4];
    for (int r = 0; r < 4; ++r) {
        CK(hipMemsetAsync(clk, 0xff, 8, st));
        CK(hipMemsetAsync(clk + 1, 0, 8, st));
        hipLaunchKernelGGL(pair_k, dim3(256), dim3(256), 0, st, (unsigned char *)a, (unsigned char *)b, part, steps, clk);
        unsigned long long h[2];
        CK(hipMemcpyAsync(h, clk, 16, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        us[r] = (double)(h[1] - h[0]) * 0.01;
    }
    std::sort(us + 1, us + 4);
    const double wrote = 16.0 * steps * 1024.0 * 1024.0;
    return wrote / us[2] * 1e-3;   // GB/s
}

// ---- the launch's traffic: readers + writers ----
struct PassArgs {
    const unsigned char *chars;
    uint64_t stride;              // input row bytes per string (multiple of 16)
    uint32_t B, M, D;
    unsigned char *plane[4];      // plane d of quad 0, string 0
    uint64_t qstep[4];            // bytes between consecutive quads of a plane
    unsigned char *masked;
    uint32_t G;                   // groups per workgroup
    uint32_t K;                   // storing waves per group: each takes the tiles t % K == k
    uint32_t per_plane;           // 1: K = D + 1 waves per group, wave d stores plane d, wave D the masked rows (the def-parallel kernel's dealing)
    uint32_t *sink;
};

__global__ __launch_bounds__(1024) void pass_k(const PassArgs a) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t n_groups = (a.B + 63u) / 64u;
    const uint32_t q4 = (a.M + 3u) / 4u, q8 = (a.M + 7u) / 8u;
    const uint32_t nwr = a.G * a.K;
    const bool reader = wave >= nwr;
    const uint32_t lg = reader ? wave - nwr : wave / a.K, k = reader ? 0u : wave % a.K;
    const size_t nb = a.B;
    for (uint32_t g = blockIdx.x + lg * gridDim.x; g < n_groups; g += gridDim.x * a.G) {
        const uint32_t b = min(g * 64u + lane, a.B - 1u);
        if (reader) {
            const unsigned char *cp = a.chars + (size_t)b * 16u;
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
        unsigned char *mp = a.masked + (size_t)b * 16u;
        if (a.per_plane) {
            if (k < a.D) {
                unsigned char *p = a.plane[k] + (size_t)b * 16u;
                const size_t qs = a.qstep[k];
                for (uint32_t q = 0; q < q4; ++q) store16_nt(p + (size_t)q * qs, make_uint4(q, 1, 2, 3));
            } else {
                for (uint32_t o = 0; o < q8; ++o) store16_nt(mp + (size_t)o * nb * 16u, make_uint4(0, 0, 0, o));
            }
            continue;
        }
        for (uint32_t q = 0; q < q4; ++q) {
            if (((q >> 4) % a.K) != k) continue;
            const uint4 v = make_uint4(q, 1, 2, 3);
            for (uint32_t d = 0; d < a.D; ++d) store16_nt(a.plane[d] + (size_t)b * 16u + (size_t)q * a.qstep[d], v);
            if ((q & 1u) && (q >> 1) < q8) store16_nt(mp + (size_t)(q >> 1) * nb * 16u, make_uint4(0, 0, 0, q));
        }
    }
}

struct Timer {
    hipEvent_t e0, e1;
    Timer() { CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); }
};

static double run_pass(const PassArgs &a, int reps, hipStream_t st, Timer &tm) {
    const uint32_t n_groups = (a.B + 63u) / 64u;
    const uint32_t grid = std::min<uint32_t>(256u, (n_groups + a.G - 1u) / a.G);
    const uint32_t waves = a.G * a.K + a.G;
    std::vector<float> ms;
    for (int r = 0; r <= reps; ++r) {
        CK(hipEventRecord(tm.e0, st));
        hipLaunchKernelGGL(pass_k, dim3(grid), dim3(64 * waves), 0, st, a);
        CK(hipEventRecord(tm.e1, st));
        CK(hipEventSynchronize(tm.e1));
        float t;
        CK(hipEventElapsedTime(&t, tm.e0, tm.e1));
        if (r) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main(int argc, char **argv) {
    size_t B = 32768, M = 32768, D = 3, cands = 20, mcands = 8, draws = 60;
    int reps = 5;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string k = argv[i];
        const size_t v = strtoull(argv[i + 1], nullptr, 10);
        if (k == "--B") B = v; else if (k == "--M") M = v; else if (k == "--D") D = v; else if (k == "--cands") cands = v;
        else if (k == "--mcands") mcands = v; else if (k == "--draws") draws = v; else if (k == "--reps") reps = (int)v;
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    if (D > 4 || D < 1) return 2;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    Timer tm;
    const size_t plane_bytes = B * M * 4, msk_bytes = B * M * 2, in_bytes = B * M;
    const double alg = (double)B * M * (1 + 4 * D + 2);
    size_t fr = 0, tot = 0;
    CK(hipMemGetInfo(&fr, &tot));
    printf("plane_probe B=%zu M=%zu D=%zu  plane %.2f GiB  masked %.2f GiB  algorithmic %.3f GB  free %.1f GiB\n", B, M, D, plane_bytes / 1073741824.0,
           msk_bytes / 1073741824.0, alg * 1e-9, fr / 1073741824.0);
    unsigned char *chars, *R;
    unsigned long long *clk;
    uint32_t *sink;
    CK(hipMalloc(&chars, in_bytes));
    CK(hipMalloc(&clk, 64));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(chars, 1, in_bytes));
    CK(hipMalloc(&R, plane_bytes * D));
    std::vector<unsigned char *> P(cands), Mk(mcands);
    for (size_t i = 0; i < cands; ++i) CK(hipMalloc(&P[i], plane_bytes));
    for (size_t i = 0; i < mcands; ++i) CK(hipMalloc(&Mk[i], msk_bytes));
    printf("R %p\n", (void *)R);
    for (size_t i = 0; i < cands; ++i) printf("P[%zu] %p\n", i, (void *)P[i]);
    for (size_t i = 0; i < mcands; ++i) printf("Mk[%zu] %p\n", i, (void *)Mk[i]);

    // ---- 2. pair matrix over the plane candidates (+ the masked candidates against them) ----
    std::vector<std::vector<double>> pm(cands, std::vector<double>(cands, 0.0));
    printf("\npair matrix, plane candidates (two equal write streams, GB/s / 100)\n     ");
    for (size_t j = 0; j < cands; ++j) printf("%4zu", j);
    printf("\n");
    for (size_t i = 0; i < cands; ++i) {
        printf("%4zu:", i);
        for (size_t j = 0; j < cands; ++j) {
            if (j < i) pm[i][j] = pm[j][i];
            else if (j > i) pm[i][j] = pair_gbs(P[i], P[j], plane_bytes, clk, st);
            printf("%4.0f", pm[i][j] / 100.0);
        }
        printf("\n");
    }
    std::vector<std::vector<double>> mm(mcands, std::vector<double>(cands, 0.0));
    printf("masked candidates (rows) against plane candidates (columns)\n");
    for (size_t i = 0; i < mcands; ++i) {
        printf("  m%zu:", i);
        for (size_t j = 0; j < cands; ++j) {
            mm[i][j] = pair_gbs(Mk[i], P[j], msk_bytes, clk, st);
            printf("%4.0f", mm[i][j] / 100.0);
        }
        printf("\n");
    }
    printf("R's thirds against each other: ");
    for (size_t i = 0; i < D; ++i)
        for (size_t j = i + 1; j < D; ++j) printf(" %zu-%zu %.0f", i, j, pair_gbs(R + i * plane_bytes, R + j * plane_bytes, plane_bytes, clk, st) / 100.0);
    printf("\n");

    auto base_args = [&](size_t m) {
        PassArgs a{};
        a.chars = chars;
        a.stride = (m + 15) / 16 * 16;
        a.B = (uint32_t)B; a.M = (uint32_t)m; a.D = (uint32_t)D;
        a.G = 2; a.K = 1; a.per_plane = 0; a.sink = sink;
        return a;
    };
    auto interleaved = [&](PassArgs &a) {
        for (size_t d = 0; d < D; ++d) { a.plane[d] = R + d * B * 16; a.qstep[d] = D * B * 16; }
    };
    auto report = [&](const char *tag, double ms, size_t m) {
        const double bytes = (double)B * m * (1 + 4 * D + 2);
        printf("%-64s %8.4f ms  %6.3f TB/s  frac %.3f\n", tag, ms, bytes / ms * 1e-9, bytes / ms * 1e-9 / 8.0);
        fflush(stdout);
    };

    // ---- 3a. today's layout against every masked candidate ----
    printf("\n== interleaved records R + masked candidate k (one storing wave per group, 2 groups per CU) ==\n");
    for (size_t k = 0; k < mcands; ++k) {
        PassArgs a = base_args(M);
        interleaved(a);
        a.masked = Mk[k];
        char tag[128];
        snprintf(tag, sizeof tag, "interleaved R + Mk[%zu]", k);
        report(tag, run_pass(a, reps, st, tm), M);
    }
    // ---- 3b. separate planes: random draws ----
    printf("\n== separate planes: random draws (p0 p1 p2 | masked), pair rates of the three planes among each other and of the masked rows against them ==\n");
    std::mt19937 rng(12345);
    struct Draw { size_t p[4], m; double ms, pr; };
    std::vector<Draw> all;
    auto pair_score = [&](const Draw &d) {
        double s = 0; int n = 0;
        for (size_t i = 0; i < D; ++i) for (size_t j = i + 1; j < D; ++j) { s += pm[d.p[i]][d.p[j]]; ++n; }
        for (size_t i = 0; i < D; ++i) { s += mm[d.m][d.p[i]]; ++n; }
        return s / n;
    };
    auto time_draw = [&](Draw &d, uint32_t per_plane) {
        PassArgs a = base_args(M);
        for (size_t i = 0; i < D; ++i) { a.plane[i] = P[d.p[i]]; a.qstep[i] = B * 16; }
        a.masked = Mk[d.m];
        if (per_plane) { a.per_plane = 1; a.K = (uint32_t)D + 1; }
        return run_pass(a, reps, st, tm);
    };
    for (size_t t = 0; t < draws; ++t) {
        Draw d{};
        std::vector<size_t> idx(cands);
        for (size_t i = 0; i < cands; ++i) idx[i] = i;
        std::shuffle(idx.begin(), idx.end(), rng);
        for (size_t i = 0; i < D; ++i) d.p[i] = idx[i];
        d.m = rng() % mcands;
        d.pr = pair_score(d);
        d.ms = time_draw(d, 0);
        all.push_back(d);
        char tag[160];
        snprintf(tag, sizeof tag, "planes %2zu %2zu %2zu | m%zu  mean pair rate %.0f", d.p[0], d.p[1], D > 2 ? d.p[2] : 0, d.m, d.pr);
        report(tag, d.ms, M);
    }
    // ---- 3c. exhaustive extremes by pair score: lowest and highest ----
    if (D == 3) {
        std::vector<Draw> ext;
        for (size_t i = 0; i < cands; ++i)
            for (size_t j = i + 1; j < cands; ++j)
                for (size_t k = j + 1; k < cands; ++k)
                    for (size_t m = 0; m < mcands; ++m) {
                        Draw d{};
                        d.p[0] = i; d.p[1] = j; d.p[2] = k; d.m = m;
                        d.pr = pair_score(d);
                        ext.push_back(d);
                    }
        std::sort(ext.begin(), ext.end(), [](const Draw &x, const Draw &y) { return x.pr < y.pr; });
        printf("\n== the 6 draws of LOWEST mean pair rate (all streams collide) and the 10 of HIGHEST (spread over the classes) ==\n");
        for (size_t t = 0; t < ext.size(); ++t) {
            if (!(t < 6 || t + 10 >= ext.size())) continue;
            Draw &d = ext[t];
            d.ms = time_draw(d, 0);
            char tag[160];
            snprintf(tag, sizeof tag, "planes %2zu %2zu %2zu | m%zu  mean pair rate %.0f", d.p[0], d.p[1], D > 2 ? d.p[2] : 0, d.m, d.pr);
            report(tag, d.ms, M);
            const double ms2 = time_draw(d, 1);
            snprintf(tag, sizeof tag, "   ... one storing wave per plane + one for the masked rows");
            report(tag, ms2, M);
        }
    }
    // ---- 4. storing waves per CU, interleaved layout, the best masked candidate of 3a would need a second pass: use Mk[0] and Mk[mcands-1] ----
    printf("\n== interleaved R: storing waves per group K (x 2 groups per CU), rows M and M / 4 ==\n");
    for (size_t m : {M, M / 4}) {
        for (uint32_t K : {1u, 2u, 4u}) {
            for (size_t mk : {(size_t)0, mcands - 1}) {
                PassArgs a = base_args(m);
                for (size_t d = 0; d < D; ++d) { a.plane[d] = R + d * B * 16; a.qstep[d] = D * B * 16; }
                a.masked = Mk[mk];
                a.K = K;
                char tag[128];
                snprintf(tag, sizeof tag, "rows %zu  K=%u (%u storing waves per CU)  Mk[%zu]", m, K, 2 * K, mk);
                report(tag, run_pass(a, reps, st, tm), m);
            }
        }
        PassArgs a = base_args(m);
        for (size_t d = 0; d < D; ++d) { a.plane[d] = R + d * B * 16; a.qstep[d] = D * B * 16; }
        a.masked = Mk[0];
        a.per_plane = 1; a.K = (uint32_t)D + 1;
        char tag[128];
        snprintf(tag, sizeof tag, "rows %zu  one storing wave per plane + masked (%zu per CU)  Mk[0]", m, 2 * (D + 1));
        report(tag, run_pass(a, reps, st, tm), m);
    }
    return 0;
}
