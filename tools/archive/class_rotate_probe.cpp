// class_rotate_probe — buffers whose consecutive 2-MiB chunks alternate between two MEASURED classes of the physical address
// space, against plain hipMalloc buffers and the library's placed pair: launch time of a position-major batch through the
// C ABI.  (tools only; gfx950; DESIGN.md §4.3 "next")
//   input: a batch dumped by tools/dump_batch.py (chars position-major, lens), regex2+regex3 or regex1
// build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/class_rotate_probe.cpp -Lhalo2_regex_amd/csrc -lhrx -Wl,-rpath,'$ORIGIN/../halo2_regex_amd/csrc' -o tools/class_rotate_probe
// (hrx_alloc_outputs_position_major is the shipped pair search)
// usage: class_rotate_probe <dfa dir> <batch file prefix> <B> <M> <regex23|regex1>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include "../include/hrx.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define HK(x) do { int r_ = (x); if (r_ != HRX_OK) { fprintf(stderr, "%s:%d %s: %d %s\n", __FILE__, __LINE__, #x, r_, hrx_last_error()); exit(1); } } while (0)
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
static const size_t CH = 2u << 20;
static hipMemAllocationProp prop;
static hipMemAccessDesc acc;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ __launch_bounds__(256) void fill2(unsigned char *a, unsigned char *b, size_t bytes) {   // 512 waves on each place
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    unsigned char *p = ((wave & 1u) ? b : a) + ((size_t)(wave >> 1) << 10) + lane * 16;
    const v4u32 v = {1, 2, 3, 4};
    for (size_t off = 0; off + (512u << 10) <= bytes; off += 512u << 10) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p + off), "v"(v) : "memory");
}
static double pair_tbs(void *a, void *b, size_t bytes) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best = 1e30;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(fill2, dim3(256), dim3(256), 0, 0, (unsigned char *)a, (unsigned char *)b, bytes); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r) best = std::min(best, (double)ms);
    }
    return 2.0 * bytes / best / 1e9;
}
struct Pool { std::vector<hipMemGenericAllocationHandle_t> h; size_t next = 0; };
static void create(Pool &p, size_t n) { for (size_t i = 0; i < n; ++i) { hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, CH, &prop, 0)); p.h.push_back(h); } }
// a virtual range of `bytes` whose chunk c comes from pools[c % npools]
static char *build(size_t bytes, Pool *pools, int npools) {
    const size_t n = (bytes + CH - 1) / CH;
    char *va; CK(hipMemAddressReserve((void **)&va, n * CH, 0, nullptr, 0));
    for (size_t c = 0; c < n; ++c) { Pool &p = pools[c % npools]; if (p.next >= p.h.size()) { fprintf(stderr, "pool exhausted\n"); exit(1); } CK(hipMemMap(va + c * CH, CH, 0, p.h[p.next++], 0)); }
    CK(hipMemSetAccess(va, n * CH, &acc, 1));
    return va;
}
int main(int argc, char **argv) {
    if (argc < 6) return 2;
    const std::string dir = argv[1], pre = argv[2]; const size_t B = atol(argv[3]), M = atol(argv[4]); const bool d2 = !strcmp(argv[5], "regex23");
    prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    hrx_defs *defs; HK(hrx_defs_create(&defs));
    for (int k : (d2 ? std::vector<int>{2, 3} : std::vector<int>{1})) {
        HK(hrx_defs_push_allstr_file(defs, (dir + "/regex" + std::to_string(k) + "_test_lookup.txt").c_str()));
        HK(hrx_defs_push_substr_file(defs, (dir + "/substr" + std::to_string(k) + "_test_lookup.txt").c_str()));
    }
    HK(hrx_defs_finalize(defs));
    hrx_ctx *ctx; HK(hrx_ctx_create(defs, 0, &ctx));
    const size_t D = d2 ? 2 : 1;
    size_t nr, nm; hrx_position_major_sizes(B, M, D, &nr, &nm);
    const size_t RB = nr * 4, MB = nm * 2, CB = B * M;
    std::vector<uint8_t> hc(CB); std::vector<uint32_t> hl(B);
    FILE *f = fopen((pre + ".chars").c_str(), "rb"); if (!f || fread(hc.data(), 1, CB, f) != CB) { fprintf(stderr, "no batch\n"); return 3; } fclose(f);
    f = fopen((pre + ".lens").c_str(), "rb"); if (!f || fread(hl.data(), 4, B, f) != B) { fprintf(stderr, "no lens\n"); return 3; } fclose(f);
    uint32_t *d_lens; uint64_t *d_st; CK(hipMalloc(&d_lens, 4 * B)); CK(hipMalloc(&d_st, 8 * B)); CK(hipMemcpy(d_lens, hl.data(), 4 * B, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](void *in, void *rec, void *msk) {
        const int lay = HRX_LAYOUT_POSITION_MAJOR | HRX_LAYOUT_INPUT_POSITION_MAJOR;
        for (int i = 0; i < 3; ++i) HK(hrx_witness_batch_device_layout(ctx, lay, (const uint8_t *)in, M, d_lens, B, M, (uint32_t *)rec, (uint16_t *)msk, d_st, nullptr));
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 12; ++i) HK(hrx_witness_batch_device_layout(ctx, lay, (const uint8_t *)in, M, d_lens, B, M, (uint32_t *)rec, (uint16_t *)msk, d_st, nullptr));
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e3 / 12;
    };
    // 1. plain
    void *in0, *rec0, *msk0; CK(hipMalloc(&in0, CB)); CK(hipMalloc(&rec0, RB)); CK(hipMalloc(&msk0, MB)); CK(hipMemcpy(in0, hc.data(), CB, hipMemcpyHostToDevice));
    printf("plain hipMalloc buffers:                 %8.1f us\n", timeit(in0, rec0, msk0));
    // 2. the library's placed pair
    uint32_t *rec1; uint16_t *msk1; HK(hrx_alloc_outputs_position_major(ctx, B, M, &rec1, &msk1));
    printf("hrx_alloc_outputs_position_major:        %8.1f us\n", timeit(in0, rec1, msk1));
    HK(hrx_device_free(rec1)); HK(hrx_device_free(msk1)); CK(hipFree(rec0)); CK(hipFree(msk0));
    // 3. two measured classes: pool A here; candidates of 1 GiB walk down until one does not collide with pool A; pool B takes its place
    double t0 = now();
    const size_t total = (RB + MB + CB + 3 * CH), half = (total / 2 + CH - 1) / CH + 8;
    Pool A, Bp; create(A, half);
    char *probeA; CK(hipMemAddressReserve((void **)&probeA, (size_t)256 << 20, 0, nullptr, 0));   // the first 256 MiB of pool A, mapped for the probe
    for (size_t c = 0; c < 128; ++c) CK(hipMemMap(probeA + c * CH, CH, 0, A.h[c], 0));
    CK(hipMemSetAccess(probeA, (size_t)256 << 20, &acc, 1));
    std::vector<void *> cands; int found = -1; double tb_same = 0, tb_found = 0;
    for (int k = 0; k < 64; ++k) {
        void *c; if (hipMalloc(&c, (size_t)1 << 30) != hipSuccess) break;
        cands.push_back(c);
        const double tb = pair_tbs(probeA, c, (size_t)256 << 20);
        if (k == 0) tb_same = tb;
        if (tb >= 6.9) { found = k; tb_found = tb; break; }
    }
    CK(hipMemUnmap(probeA, (size_t)256 << 20));
    if (found >= 0) { CK(hipFree(cands[found])); cands[found] = nullptr; }
    // pool B: from where the good candidate was (its GiB first, then onwards below the held candidates)
    create(Bp, half);
    for (void *c : cands) if (c) CK(hipFree(c));
    printf("two-class pools: candidate %d of %zu did not collide (%.2f TB/s; the first: %.2f); %.2f s\n", found, cands.size(), tb_found, tb_same, now() - t0);
    Pool pools[2] = {A, Bp};
    char *rec2 = build(RB, pools, 2), *msk2 = build(MB, pools, 2), *in2 = build(CB, pools, 2);
    CK(hipMemcpy(in2, hc.data(), CB, hipMemcpyHostToDevice));
    printf("all three alternate A/B per 2 MiB:       %8.1f us   (%.2f s to build)\n", timeit(in2, rec2, msk2), now() - t0);
    printf("records and masked rows alternate, plain input: %8.1f us\n", timeit(in0, rec2, msk2));
    return 0;
}
