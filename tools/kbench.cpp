// kbench.cpp — on-box kernel micro-harness (profiling only; not part of the product or the tests).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kbench.cpp halo2_regex_amd/csrc/hrx_defs.cpp \
//         halo2_regex_amd/csrc/hrx_kernel.hip halo2_regex_amd/csrc/hrx_kernel_pm.hip halo2_regex_amd/csrc/hrx_kernel_sm.hip -o tools/kbench
//   tools/kbench <allstr.txt> <substr.txt> [B] [n] [M] [steps] [debug] [stamps]
// Runs the witness kernel on alphabet-uniform noise and prints the average launch time; with stamps=1 it also
// dumps the per-tile s_memtime stamps of a few waves (walk / epilogue / store phases).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../halo2_regex_amd/csrc/hrx_defs.hpp"
#include "../halo2_regex_amd/csrc/hrx_kernel.hpp"

using namespace hrx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static std::string slurp(const char *p) { std::ifstream f(p, std::ios::binary); std::ostringstream s; s << f.rdbuf(); return s.str(); }

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage\n"); return 2; }
    const size_t B = argc > 3 ? atol(argv[3]) : 65536, n = argc > 4 ? atol(argv[4]) : 1023, M = argc > 5 ? atol(argv[5]) : 1024;
    const int steps = argc > 6 ? atoi(argv[6]) : 50;
    const uint32_t debug = argc > 7 ? (uint32_t)strtoul(argv[7], 0, 0) : 0;
    const int want_stamps = argc > 8 ? atoi(argv[8]) : 0;
    const size_t rec_off = argc > 9 ? atol(argv[9]) : 0, msk_off = argc > 10 ? atol(argv[10]) : 0, chr_off = argc > 11 ? atol(argv[11]) : 0;
    const size_t rec_pad = argc > 12 ? atol(argv[12]) : 0, msk_pad = argc > 13 ? atol(argv[13]) : 0, chr_pad = argc > 14 ? atol(argv[14]) : 0;
    DefsSet s;
    RegexDefs rd;
    std::string t = slurp(argv[1]);
    if (parse_allstr_text(t.data(), t.size(), rd.allstr)) return 3;
    SubstrRegexDef sd;
    t = slurp(argv[2]);
    if (parse_substr_text(t.data(), t.size(), sd)) return 3;
    rd.substrs.push_back(sd);
    s.defs.push_back(rd);
    std::string err;
    if (finalize_defs(s, err)) { fprintf(stderr, "%s\n", err.c_str()); return 4; }
    const size_t stride = (n + 15) / 16 * 16 + chr_pad;
    std::vector<uint8_t> h(B * stride);
    uint64_t x = 88172645463325252ull;
    static const uint8_t alpha[98] = {9, 10, 13};
    std::vector<uint8_t> al(98);
    for (int i = 0; i < 98; ++i) al[i] = i < 3 ? alpha[i] : (uint8_t)(32 + i - 3);
    for (auto &c : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; c = al[x % 98]; }
    std::vector<uint32_t> lens(B, (uint32_t)n);
    uint8_t *d_chars; uint32_t *d_lens, *d_rec, *d_tab; uint16_t *d_msk; uint64_t *d_st; unsigned long long *d_stamps = nullptr;
    CK(hipMalloc(&d_chars, h.size() + (1 << 22))); CK(hipMalloc(&d_lens, 4 * B)); CK(hipMalloc(&d_rec, 4 * B * (M + rec_pad) + (1 << 22))); CK(hipMalloc(&d_msk, 2 * B * (M + msk_pad) + (1 << 22)));
    d_chars += chr_off; d_rec += rec_off / 4; d_msk += msk_off / 2;
    printf("bases: chars %p rec %p msk %p\n", (void *)d_chars, (void *)d_rec, (void *)d_msk);
    CK(hipMalloc(&d_st, 8 * B)); CK(hipMalloc(&d_tab, s.table_image.size() * 4));
    if ((getenv("KB_LAYOUT") ? atoi(getenv("KB_LAYOUT")) : 0) & 2) {  // position-major input: [stride/16][B][16]
        std::vector<uint8_t> t2(h.size());
        for (size_t bb = 0; bb < B; ++bb) for (size_t i = 0; i < stride; ++i) t2[((i / 16) * B + bb) * 16 + i % 16] = h[bb * stride + i];
        h.swap(t2);
    }
    if ((getenv("KB_LAYOUT") ? atoi(getenv("KB_LAYOUT")) : 0) & 4) {  // group-major input: [B/64][stride/16][64][16]
        std::vector<uint8_t> t2(h.size());
        for (size_t bb = 0; bb < B; ++bb) for (size_t i = 0; i < stride; ++i) t2[(((bb / 64) * (stride / 16) + i / 16) * 64 + bb % 64) * 16 + i % 16] = h[bb * stride + i];
        h.swap(t2);
    }
    CK(hipMemcpy(d_chars, h.data(), h.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(d_lens, lens.data(), 4 * B, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tab, s.table_image.data(), s.table_image.size() * 4, hipMemcpyHostToDevice));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    WitnessArgs a{};
    a.chars = d_chars; a.stride = stride; a.lens = d_lens; a.B = (uint32_t)B; a.M = (uint32_t)M;
    a.rec_pitch = (uint32_t)(M + rec_pad); a.msk_pitch = (uint32_t)(M + msk_pad);
    a.layout = getenv("KB_LAYOUT") ? atoi(getenv("KB_LAYOUT")) : 0; a.records = d_rec; a.masked = d_msk;
    a.status = d_st; a.table_image = d_tab; a.table_bytes = (uint32_t)(s.table_image.size() * 4);
    a.D = 1; a.debug = debug; a.dc[0] = s.consts[0];
    LaunchInfo li;
    if (!plan_witness_launch(a, prop.multiProcessorCount, li)) return 5;
    const size_t ntiles = (M + 63) / 64, nw = (size_t)li.grid * li.waves_per_wg;
    if (want_stamps) { CK(hipMalloc(&d_stamps, nw * ntiles * 128)); CK(hipMemset(d_stamps, 0, nw * ntiles * 128)); }
    printf("grid %d x %d waves, gs %u, split %d, lds %zu, clock %d kHz\n", li.grid, li.waves_per_wg, a.gs, li.split, li.lds_bytes, prop.clockRate);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) CK(launch_witness(a, li, 0));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < steps; ++i) CK(launch_witness(a, li, 0));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / steps;
    printf("debug=0x%x  %.2f us/launch  %.3e rows/s  %.1f%% of 8 TB/s\n", debug, us, (double)B * n / (us * 1e-6), 7.0 * B * n / (us * 1e-6) / 8e12 * 100);
    if (want_stamps && li.split == 2) {   // position-major kernel built with -DHRX_STAMPS: per walker [input wait, walk, tile end, whole group] ticks
        CK(hipMemset(d_stamps, 0, nw * ntiles * 128));
        a.stamps = d_stamps;
        const int reps = 20;
        const size_t units = (size_t)li.grid * (((a.layout & 1u) && !li.half) ? li.waves_per_wg / 3 : li.waves_per_wg / 2);   // pairs per workgroup (walker + loader + finisher)
        for (int i = 0; i < reps; ++i) CK(launch_witness(a, li, 0));
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> st(units * 8);
        CK(hipMemcpy(st.data(), d_stamps, st.size() * 8, hipMemcpyDeviceToHost));
        double w = 0, k = 0, e = 0, g = 0, gmax = 0;
        for (size_t u = 0; u < units; ++u) { w += st[u * 8]; k += st[u * 8 + 1]; e += st[u * 8 + 2]; g += st[u * 8 + 3]; if (st[u * 8 + 3] > gmax) gmax = st[u * 8 + 3]; }
        const double per = (double)units * reps * ntiles;
        printf("position-major walker, mean s_memtime ticks per 64-row tile: input wait %.0f, walk %.0f, tile end %.0f; per launch and walker: %.0f ticks in its groups (max %.0f)\n",
               w / per, k / per, e / per, g / units / reps, gmax / reps);
        // timeline of the LAST launch on the 100-MHz wall clock: workgroup entry, walker start (after the table staging), walker end
        unsigned long long e0 = ~0ull, e1 = 0, s1 = 0, x0 = ~0ull, x1 = 0; double pro = 0;
        for (size_t u = 0; u < units; ++u) {
            const unsigned long long en = st[u * 8 + 4], sr = st[u * 8 + 5], ex = st[u * 8 + 6];
            e0 = std::min(e0, en); e1 = std::max(e1, en); s1 = std::max(s1, sr); x0 = std::min(x0, ex); x1 = std::max(x1, ex); pro += (double)(sr - en);
        }
        {   // who finishes late?  mean walker-done time by workgroup index mod 8 (the XCD a workgroup lands on, round-robin) and by pair
            const size_t pw = units / (size_t)li.grid;
            double xs[8] = {0}, ps[8] = {0}; size_t xn[8] = {0}, pn[8] = {0};
            for (size_t u = 0; u < units; ++u) {
                const double t = (double)(st[u * 8 + 6] - e0) / 100.0;
                xs[(u / pw) % 8] += t; xn[(u / pw) % 8]++; ps[u % pw] += t; pn[u % pw]++;
            }
            printf("mean walker-done us by workgroup %% 8:");
            for (int x = 0; x < 8; ++x) printf(" %.1f", xs[x] / (xn[x] ? xn[x] : 1));
            printf("; by pair:");
            for (size_t q = 0; q < pw && q < 8; ++q) printf(" %.1f", ps[q] / (pn[q] ? pn[q] : 1));
            printf("\n  per workgroup (mean of its pairs), 32 per line:\n");
            for (size_t w = 0; w < (size_t)li.grid; ++w) {
                double t = 0; for (size_t q = 0; q < pw; ++q) t += (double)(st[(w * pw + q) * 8 + 6] - e0) / 100.0;
                printf(" %.0f", t / pw); if (w % 32 == 31) printf("\n");
            }
        }
        printf("last launch, us after the first workgroup's entry: last workgroup enters %.2f, mean staging %.2f, last walker starts %.2f, first walker done %.2f, last walker done %.2f\n",
               (e1 - e0) / 100.0, pro / units / 100.0, (s1 - e0) / 100.0, (x0 - e0) / 100.0, (x1 - e0) / 100.0);
    } else if (want_stamps) {
        a.stamps = d_stamps;
        CK(launch_witness(a, li, 0)); CK(hipDeviceSynchronize());
        const int per = li.split ? 8 : 4;
        const size_t units = li.split ? (size_t)li.grid * (li.waves_per_wg / 2) : nw;
        const size_t nt = li.split ? (M + 31) / 32 : ntiles;
        std::vector<unsigned long long> st(units * nt * per);
        CK(hipMemcpy(st.data(), d_stamps, st.size() * 8, hipMemcpyDeviceToHost));
        if (!li.split) {
            for (size_t w : {(size_t)0, nw / 2}) {
                printf("wave %zu: per tile [walk, epilogue, store-issue, gap-to-next]\n", w);
                for (size_t tt = 0; tt < nt; ++tt) {
                    const unsigned long long *q = &st[(w * nt + tt) * 4];
                    const unsigned long long nxt = tt + 1 < nt ? q[4] : q[3];
                    printf("  t%02zu %6llu %6llu %6llu %6llu\n", tt, q[1] - q[0], q[2] - q[1], q[3] - q[2], nxt - q[3]);
                }
            }
        } else {
            double ww = 0, wk = 0, wr = 0, sw = 0, ss = 0; size_t cnt = 0;
            for (size_t u = 0; u < units; ++u) for (size_t tt = 0; tt < nt; ++tt) {
                const unsigned long long *q = &st[(u * nt + tt) * 8];
                ww += q[1] - q[0]; wk += q[2] - q[1]; wr += q[3] - q[2]; sw += q[5] - q[4]; ss += q[6] - q[5]; ++cnt;
            }
            printf("mean cycles per 32-row tile: walker [ring-wait %.0f, walk+masks %.0f, rotate/loads %.0f]  storer [ring-wait %.0f, move %.0f]\n",
                   ww / cnt, wk / cnt, wr / cnt, sw / cnt, ss / cnt);
            for (size_t u : {(size_t)0, units / 2}) {
                printf("pair %zu: per tile walker[wait walk rot] storer[wait move]\n", u);
                for (size_t tt = 0; tt < nt; ++tt) {
                    const unsigned long long *q = &st[(u * nt + tt) * 8];
                    printf("  t%02zu %6llu %6llu %6llu | %6llu %6llu\n", tt, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[5] - q[4], q[6] - q[5]);
                }
            }
        }
    }
    return 0;
}
