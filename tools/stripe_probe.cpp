// stripe_probe.cpp — on-box probe (round 6): would ONE def's records gain from being spread over the classes of the device memory too?  A D = 1 launch writes 4 of its 6 output bytes per
// row into the records and 2 into the masked rows (placed in another class by hrx_alloc_output_pair: two classes at 2 : 1).  Here the records are R STRIPES — quad q of a string in stripe
// q % R at slot q / R, every stripe a buffer of its own in a class of its own — and the no-compute pass of the position-major kernel (4 reader + 4 writer waves per CU, a writer stores its
// group's record quads and masked octets in row order) runs over R = 1, 2, 3, 4 with all R + 1 buffers in mutually non-colliding arenas (as far as the box has them), and once with
// everything in ONE class.  Buffers are carved out of 2-GiB arenas (hipMalloc), `sets` rotating buffer sets at different offsets of the same arenas (the Infinity Cache must not hold a set
// between two uses: the bench line's regime).
//   stripe_probe --B 65536 --M 8192 --sets 1      (cfg 5's bytes: 2 GiB of records per launch)
//   stripe_probe --B 65536 --M 1024 --sets 8      (the bench line's shape)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static __device__ __forceinline__ void store16_nt(void *p, const uint4 &v) {
    typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(p), "v"(v4u32{v.x, v.y, v.z, v.w}) : "memory");
}

__global__ __launch_bounds__(256) void pair_k(unsigned char *a, unsigned char *b, size_t part, uint32_t steps, unsigned long long *clk) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    unsigned char *base = ((wave & 1u) ? b : a) + ((size_t)(wave >> 1) << 10) + lane * 16u;
    const size_t window = (size_t)512 << 10;
    for (uint32_t k = 0; k < 16u; ++k)
        for (uint32_t s = 0; s < steps; ++s) store16_nt(base + k * part + s * window, make_uint4(0, 0, 0, 0));
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
    return 16.0 * steps * 1024.0 * 1024.0 / us[2] * 1e-3;
}

struct PassArgs {
    const unsigned char *chars;
    uint32_t B, M, R;
    unsigned char *stripe[4];
    unsigned char *masked;
    uint32_t *sink;
};

__global__ __launch_bounds__(512) void pass_k(const PassArgs a) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t n_groups = (a.B + 63u) / 64u;
    const uint32_t q4 = (a.M + 3u) / 4u, q8 = (a.M + 7u) / 8u;
    const size_t nb = a.B;
    for (uint32_t g = blockIdx.x * 4u + (wave & 3u); g < n_groups; g += gridDim.x * 4u) {
        const uint32_t b = min(g * 64u + lane, a.B - 1u);
        if (wave >= 4u) {
            const unsigned char *cp = a.chars + (size_t)b * 16u;
            uint4 acc = make_uint4(0, 0, 0, 0);
            const uint32_t nchunk = a.M / 16u;
#pragma unroll 8
            for (uint32_t c = 0; c < nchunk; ++c) {
                const uint4 v = *reinterpret_cast<const uint4 *>(cp + (size_t)c * nb * 16u);
                acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
            }
            if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) a.sink[0] = acc.z ^ acc.w;
            continue;
        }
        unsigned char *mp = a.masked + (size_t)b * 16u;
        for (uint32_t q = 0; q < q4; ++q) {
            const uint32_t s = q % a.R, slot = q / a.R;
            store16_nt(a.stripe[s] + (size_t)b * 16u + (size_t)slot * nb * 16u, make_uint4(q, 1, 2, 3));
            if ((q & 1u) && (q >> 1) < q8) store16_nt(mp + (size_t)(q >> 1) * nb * 16u, make_uint4(0, 0, 0, q));
        }
    }
}

int main(int argc, char **argv) {
    size_t B = 65536, M = 8192, sets = 1, arenas = 14, steps = 40;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string k = argv[i];
        const size_t v = strtoull(argv[i + 1], nullptr, 10);
        if (k == "--B") B = v; else if (k == "--M") M = v; else if (k == "--sets") sets = v; else if (k == "--arenas") arenas = v; else if (k == "--steps") steps = v;
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const size_t AR = (size_t)2 << 30;
    const size_t rec_bytes = B * M * 4, msk_bytes = B * M * 2, in_bytes = B * M;
    const double alg = (double)B * M * 7;
    printf("stripe_probe B=%zu M=%zu: records %.0f MiB, masked %.0f MiB, input %.0f MiB per launch; %zu buffer sets; algorithmic %.3f GB\n", B, M, rec_bytes / 1048576.0, msk_bytes / 1048576.0,
           in_bytes / 1048576.0, sets, alg * 1e-9);
    if (sets * rec_bytes > AR) { fprintf(stderr, "sets x records must fit a 2-GiB arena\n"); return 2; }
    unsigned long long *clk;
    uint32_t *sink;
    unsigned char *chars;
    CK(hipMalloc(&clk, 64));
    CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&chars, in_bytes * sets));
    CK(hipMemset(chars, 1, in_bytes * sets));
    std::vector<unsigned char *> A(arenas);
    for (size_t i = 0; i < arenas; ++i) CK(hipMalloc(&A[i], AR));
    std::vector<std::vector<double>> pm(arenas, std::vector<double>(arenas, 0.0));
    std::vector<double> all;
    printf("arena pairings (GB/s / 10)\n");
    for (size_t i = 0; i < arenas; ++i) {
        printf("%3zu:", i);
        for (size_t j = 0; j < arenas; ++j) {
            if (j > i) { pm[i][j] = pair_gbs(A[i], A[j], AR, clk, st); all.push_back(pm[i][j]); }
            else if (j < i) pm[i][j] = pm[j][i];
            printf("%5.0f", pm[i][j] / 10);
        }
        printf("\n");
    }
    double lo = *std::min_element(all.begin(), all.end()), hi = *std::max_element(all.begin(), all.end()), cut = 0.5 * (lo + hi);
    for (int it = 0; it < 8; ++it) {
        double sl = 0, sh = 0; size_t nl = 0, nh = 0;
        for (double v : all) { if (v < cut) { sl += v; ++nl; } else { sh += v; ++nh; } }
        if (!nl || !nh) break;
        cut = 0.5 * (sl / nl + sh / nh);
    }
    printf("pairings %.0f .. %.0f GB/s, cut %.0f\n", lo, hi, cut);
    // largest set of mutually non-colliding arenas (brute force over subsets, arenas <= 16)
    auto clique = [&](size_t want, bool collide) {
        std::vector<size_t> best;
        double best_sum = collide ? 1e30 : -1;
        for (uint32_t m = 0; m < (1u << arenas); ++m) {
            if ((size_t)__builtin_popcount(m) != want) continue;
            std::vector<size_t> v;
            for (size_t i = 0; i < arenas; ++i) if (m >> i & 1) v.push_back(i);
            double sum = 0; bool ok = true;
            for (size_t x = 0; x < v.size() && ok; ++x)
                for (size_t y = x + 1; y < v.size(); ++y) { const double r = pm[v[x]][v[y]]; sum += r; if (collide ? r >= cut : r < cut) { ok = false; break; } }
            if (!ok) continue;
            if (collide ? sum < best_sum : sum > best_sum) { best_sum = sum; best = v; }
        }
        return best;
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](const char *tag, size_t R, const std::vector<size_t> &ar) {      // ar: R record arenas + the masked rows' arena
        std::vector<PassArgs> pa(sets);
        const size_t stripe_bytes = (rec_bytes / R + 4095) / 4096 * 4096;
        for (size_t k = 0; k < sets; ++k) {
            PassArgs &a = pa[k];
            a.chars = chars + k * in_bytes; a.B = (uint32_t)B; a.M = (uint32_t)M; a.R = (uint32_t)R; a.sink = sink;
            for (size_t s = 0; s < R; ++s) a.stripe[s] = A[ar[s]] + k * stripe_bytes + (ar[s] == ar[R] ? sets * msk_bytes : 0);
            a.masked = A[ar[R]] + k * msk_bytes;
            for (size_t s = 0; s < R; ++s)       // two stripes in one arena (the one-class case): behind each other
                for (size_t s2 = 0; s2 < s; ++s2) if (ar[s] == ar[s2]) a.stripe[s] += sets * stripe_bytes * 1;
        }
        const uint32_t grid = (uint32_t)std::min<size_t>(256, ((B + 63) / 64 + 3) / 4);
        float best = 1e30f, sum = 0;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, st));
            for (size_t i = 0; i < steps; ++i) hipLaunchKernelGGL(pass_k, dim3(grid), dim3(512), 0, st, pa[i % sets]);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float t;
            CK(hipEventElapsedTime(&t, e0, e1));
            if (rep) { best = std::min(best, t / steps); sum += t / steps; }
        }
        std::string as;
        for (size_t s = 0; s <= R; ++s) as += (s == R ? "| m " : "") + std::to_string(ar[s]) + " ";
        printf("%-28s R=%zu arenas %-22s %8.4f ms (min %.4f)  %6.3f TB/s  frac %.3f\n", tag, R, as.c_str(), sum / 3, best, alg / (sum / 3) * 1e-9, alg / (sum / 3) * 1e-9 / 8.0);
        fflush(stdout);
    };
    for (size_t R = 1; R <= 4; ++R) {
        if (sets * ((rec_bytes / R + 4095) / 4096 * 4096) * (R > 1 ? 1 : 1) > AR) continue;
        std::vector<size_t> c = clique(R + 1, false);
        if (c.empty()) { printf("R=%zu: no %zu mutually non-colliding arenas among %zu\n", R, R + 1, arenas); continue; }
        run("spread over the classes", R, c);
        std::rotate(c.begin(), c.begin() + 1, c.end());
        run("spread (rotated)", R, c);
    }
    {   // everything in one class: R = 1 with colliding arenas, and both buffers in ONE arena
        std::vector<size_t> c = clique(2, true);
        if (!c.empty()) run("records | masked collide", 1, c);
        if (sets * (rec_bytes + msk_bytes) <= AR) run("one arena", 1, {0, 0});
        std::vector<size_t> c3 = clique(3, true);
        if (!c3.empty()) run("R=2, all three collide", 2, c3);
    }
    return 0;
}
