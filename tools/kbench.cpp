// kbench.cpp — on-box kernel micro-harness (profiling only; not part of the product or the tests).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kbench.cpp halo2_regex_amd/csrc/hrx_defs.cpp \
//         halo2_regex_amd/csrc/hrx_kernel.hip -o tools/kbench
//   tools/kbench <allstr.txt> <substr.txt> [B] [n] [M] [steps] [debug] [stamps]
// Runs the witness kernel on alphabet-uniform noise and prints the average launch time; with stamps=1 it also
// dumps the per-tile s_memtime stamps of a few waves (walk / epilogue / store phases).
#include <hip/hip_runtime.h>

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
    const size_t stride = (n + 15) / 16 * 16;
    std::vector<uint8_t> h(B * stride);
    uint64_t x = 88172645463325252ull;
    static const uint8_t alpha[98] = {9, 10, 13};
    std::vector<uint8_t> al(98);
    for (int i = 0; i < 98; ++i) al[i] = i < 3 ? alpha[i] : (uint8_t)(32 + i - 3);
    for (auto &c : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; c = al[x % 98]; }
    std::vector<uint32_t> lens(B, (uint32_t)n);
    uint8_t *d_chars; uint32_t *d_lens, *d_rec, *d_tab; uint16_t *d_msk; uint64_t *d_st; unsigned long long *d_stamps = nullptr;
    CK(hipMalloc(&d_chars, h.size())); CK(hipMalloc(&d_lens, 4 * B)); CK(hipMalloc(&d_rec, 4 * B * M)); CK(hipMalloc(&d_msk, 2 * B * M));
    CK(hipMalloc(&d_st, 8 * B)); CK(hipMalloc(&d_tab, s.table_image.size() * 4));
    CK(hipMemcpy(d_chars, h.data(), h.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(d_lens, lens.data(), 4 * B, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tab, s.table_image.data(), s.table_image.size() * 4, hipMemcpyHostToDevice));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    WitnessArgs a{};
    a.chars = d_chars; a.stride = stride; a.lens = d_lens; a.B = (uint32_t)B; a.M = (uint32_t)M; a.records = d_rec; a.masked = d_msk;
    a.status = d_st; a.table_image = d_tab; a.table_bytes = (uint32_t)(s.table_image.size() * 4);
    a.D = 1; a.debug = debug; a.dc[0] = s.consts[0];
    LaunchInfo li;
    if (!plan_witness_launch(a, prop.multiProcessorCount, li)) return 5;
    const size_t ntiles = (M + 63) / 64, nw = (size_t)li.grid * li.waves_per_wg;
    if (want_stamps) { CK(hipMalloc(&d_stamps, nw * ntiles * 32)); CK(hipMemset(d_stamps, 0, nw * ntiles * 32)); }
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
    if (want_stamps) {
        a.stamps = d_stamps;
        CK(launch_witness(a, li, 0)); CK(hipDeviceSynchronize());
        std::vector<unsigned long long> st(nw * ntiles * 4);
        CK(hipMemcpy(st.data(), d_stamps, st.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t00 = ~0ull;
        for (size_t w = 0; w < nw; ++w) if (st[w * ntiles * 4] && st[w * ntiles * 4] < t00) t00 = st[w * ntiles * 4];
        for (size_t w : {(size_t)0, (size_t)1, nw / 2, nw - 1}) {
            printf("wave %zu: start+%llu; per tile [walk, epilogue, store-issue, gap-to-next]\n", w, st[w * ntiles * 4] - t00);
            for (size_t tt = 0; tt < ntiles; ++tt) {
                const unsigned long long *q = &st[(w * ntiles + tt) * 4];
                const unsigned long long nxt = tt + 1 < ntiles ? q[4] : q[3];
                printf("  t%02zu %6llu %6llu %6llu %6llu\n", tt, q[1] - q[0], q[2] - q[1], q[3] - q[2], nxt - q[3]);
            }
        }
        double sw = 0, se = 0, ss = 0; size_t cnt = 0;
        unsigned long long tend = 0;
        for (size_t w = 0; w < nw; ++w) for (size_t tt = 0; tt < ntiles; ++tt) { const unsigned long long *q = &st[(w * ntiles + tt) * 4]; sw += q[1] - q[0]; se += q[2] - q[1]; ss += q[3] - q[2]; ++cnt; if (q[3] > tend) tend = q[3]; }
        printf("mean cycles per tile: walk %.0f  epilogue %.0f  store %.0f ; kernel span %llu ticks (s_memtime)\n", sw / cnt, se / cnt, ss / cnt, tend - t00);
    }
    return 0;
}
