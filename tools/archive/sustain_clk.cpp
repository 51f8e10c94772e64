// sustain_clk.cpp — on-box probe (round 2): does the shader clock hold while the bench-line kernel runs back to back?
// Links libhrx.so through the C ABI only.  Launches the witness kernel (regex1 + substr1, 65536 x 1023 B, position-major)
// in chunks of CH launches on one stream; between chunks a one-wave probe kernel runs a fixed dependent VALU chain and
// reads s_memtime / s_memrealtime (100 MHz) around it: chain steps per microsecond = a direct reading of the shader clock
// at that moment.  Prints per chunk: us per witness launch, probe MHz-equivalent.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/sustain_clk.cpp -Iinclude -Lhalo2_regex_amd/csrc -lhrx -Wl,-rpath,$PWD/halo2_regex_amd/csrc -o tools/sustain_clk
//   tools/sustain_clk [chunks] [launches per chunk] [gap_us between chunks] [spin: extra busy waves during the run]
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "hrx.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define HK(x) do { int r_ = (x); if (r_) { fprintf(stderr, "%s: %d %s\n", #x, r_, hrx_last_error()); exit(1); } } while (0)

constexpr int CHAIN = 100000;
__global__ void probe_k(unsigned long long *out) {
    unsigned v = threadIdx.x;
    const unsigned long long w0 = wall_clock64(), t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < CHAIN / 50; ++i) {
#pragma unroll
        for (int k = 0; k < 50; ++k) asm volatile("v_add_u32 %0, %0, 1" : "+v"(v));
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; out[2] = v; }
}

// the no-compute traffic mix of tools/mixceil.cpp (pair, non-temporal stores): 4 reader + 4 writer waves per CU
__global__ __launch_bounds__(512) void pairnt_k(const uint4 *__restrict__ in, uint4 *__restrict__ rec, uint4 *__restrict__ msk, unsigned *sink) {
    constexpr size_t B = 65536, M = 1024;
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
        __builtin_nontemporal_store(v4{(unsigned)q, 1, 2, 3}, reinterpret_cast<v4 *>(rec + q * B + b));
        if (q & 1) __builtin_nontemporal_store(v4{0, 0, 0, (unsigned)q}, reinterpret_cast<v4 *>(msk + (q >> 1) * B + b));
    }
}

int main(int argc, char **argv) {
    const int load = argc > 4 ? atoi(argv[4]) : 0;   // 0: the witness kernel; 1: the no-compute traffic mix
    const int chunks = argc > 1 ? atoi(argv[1]) : 40, CH = argc > 2 ? atoi(argv[2]) : 40, gap_us = argc > 3 ? atoi(argv[3]) : 0;
    const size_t B = 65536, n = 1023, M = 1024, stride = 1024;
    std::string root = getenv("GRAFT_REPO_ROOT") ? getenv("GRAFT_REPO_ROOT") : "/root/repo";
    hrx_defs *defs; HK(hrx_defs_create(&defs));
    HK(hrx_defs_push_allstr_file(defs, (root + "/tests/golden/dfa/regex1_test_lookup.txt").c_str()));
    HK(hrx_defs_push_substr_file(defs, (root + "/tests/golden/dfa/substr1_test_lookup.txt").c_str()));
    HK(hrx_defs_finalize(defs));
    hrx_ctx *ctx; HK(hrx_ctx_create(defs, 0, &ctx));
    std::vector<uint8_t> h(B * stride);
    uint64_t x = 88172645463325252ull;
    for (auto &c : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; const unsigned k = x % 98; c = k < 3 ? (k == 0 ? 9 : k == 1 ? 10 : 13) : (uint8_t)(32 + k - 3); }
    std::vector<uint32_t> lens(B, (uint32_t)n);
    uint8_t *d_chars; uint32_t *d_lens, *d_rec; uint16_t *d_msk; uint64_t *d_st; unsigned long long *d_p;
    CK(hipMalloc(&d_chars, h.size())); CK(hipMalloc(&d_lens, 4 * B)); CK(hipMalloc(&d_rec, 4 * B * M)); CK(hipMalloc(&d_msk, 2 * B * M)); CK(hipMalloc(&d_st, 8 * B));
    CK(hipMalloc(&d_p, 24 * (chunks + 1)));
    CK(hipMemcpy(d_chars, h.data(), h.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(d_lens, lens.data(), 4 * B, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    auto launch = [&] {
        if (load == 1) { hipLaunchKernelGGL(pairnt_k, dim3(256), dim3(512), 0, st, (const uint4 *)d_chars, (uint4 *)d_rec, (uint4 *)d_msk, (unsigned *)d_st); return; }
        HK(hrx_witness_batch_device_layout(ctx, HRX_LAYOUT_POSITION_MAJOR | HRX_LAYOUT_INPUT_POSITION_MAJOR, d_chars, stride, d_lens, B, M, d_rec, d_msk, d_st, st));
    };
    for (int i = 0; i < 3; ++i) launch();
    CK(hipStreamSynchronize(st));
    usleep(500000);
    std::vector<hipEvent_t> ev(2 * chunks);
    for (auto &e : ev) CK(hipEventCreate(&e));
    hipLaunchKernelGGL(probe_k, dim3(1), dim3(64), 0, st, d_p);   // the clock before the load starts
    for (int c = 0; c < chunks; ++c) {
        CK(hipEventRecord(ev[2 * c], st));
        for (int k = 0; k < CH; ++k) launch();
        CK(hipEventRecord(ev[2 * c + 1], st));
        hipLaunchKernelGGL(probe_k, dim3(1), dim3(64), 0, st, d_p + 3 * (c + 1));
        if (gap_us) { CK(hipStreamSynchronize(st)); if (gap_us > 1) usleep(gap_us); }   // gap 1: synchronize only
    }
    CK(hipStreamSynchronize(st));
    std::vector<unsigned long long> p(3 * (chunks + 1));
    CK(hipMemcpy(p.data(), d_p, p.size() * 8, hipMemcpyDeviceToHost));
    uint64_t s0; CK(hipMemcpy(&s0, d_st, 8, hipMemcpyDeviceToHost));
    printf("load %d, %d chunks x %d launches, gap %d us; status[0] = %llu; probe: %d dependent v_add_u32; columns: us/launch | chain steps per us | s_memtime ticks per us\n", load, chunks, CH, gap_us, (unsigned long long)s0, CHAIN);
    printf("before load:        | %.0f | %.0f\n", CHAIN / (p[1] / 100.0), p[0] / (p[1] / 100.0));
    for (int c = 0; c < chunks; ++c) {
        float ms; CK(hipEventElapsedTime(&ms, ev[2 * c], ev[2 * c + 1]));
        const unsigned long long *q = &p[3 * (c + 1)];
        printf("chunk %2d: %6.1f | %.0f | %.0f\n", c, ms * 1e3 / CH, CHAIN / (q[1] / 100.0), q[0] / (q[1] / 100.0));
    }
    return 0;
}
