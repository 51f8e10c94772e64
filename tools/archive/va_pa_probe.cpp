// va_pa_probe — the witness launch time goes with the allocation that holds the outputs (tools/set_probe*.py).  With the
// virtual range or with the physical pages?  Three reserved virtual ranges x three sets of physical chunks (HIP virtual memory
// management), every combination mapped in turn and timed.  regex1, 262144 x 1024 B, position-major.  (tools only)
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/va_pa_probe.cpp -Ihalo2_regex_amd/../include -Lhalo2_regex_amd/csrc -lhrx -Wl,-rpath,$PWD/halo2_regex_amd/csrc -o tools/va_pa_probe
//   tools/va_pa_probe tests/golden/dfa [chunk MiB = 16]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "../include/hrx.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define HK(x) do { int r_ = (x); if (r_ != HRX_OK) { fprintf(stderr, "%s:%d %s: %d %s\n", __FILE__, __LINE__, #x, r_, hrx_last_error()); exit(1); } } while (0)
int main(int argc, char **argv) {
    const std::string dir = argc > 1 ? argv[1] : "tests/golden/dfa";
    const size_t chunk = (size_t)(argc > 2 ? atol(argv[2]) : 16) << 20;
    const size_t B = 262144, n = 1023, M = 1024, NS = 3;
    hrx_defs *defs; HK(hrx_defs_create(&defs));
    HK(hrx_defs_push_allstr_file(defs, (dir + "/regex1_test_lookup.txt").c_str()));
    HK(hrx_defs_push_substr_file(defs, (dir + "/substr1_test_lookup.txt").c_str()));
    HK(hrx_defs_finalize(defs));
    hrx_ctx *ctx; HK(hrx_ctx_create(defs, 0, &ctx));
    size_t nr, nm; hrx_position_major_sizes(B, M, 1, &nr, &nm);
    const size_t rec_bytes = (nr * 4 + chunk - 1) / chunk * chunk, msk_bytes = (nm * 2 + chunk - 1) / chunk * chunk, tot = rec_bytes + msk_bytes;
    std::vector<uint8_t> h(B * M);
    uint64_t x = 88172645463325252ull;
    for (auto &c : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; c = (uint8_t)(32 + x % 95); }
    std::vector<uint32_t> lens(B, (uint32_t)n);
    uint8_t *d_chars; uint32_t *d_lens; uint64_t *d_st;
    CK(hipMalloc(&d_chars, h.size())); CK(hipMalloc(&d_lens, 4 * B)); CK(hipMalloc(&d_st, 8 * B));
    CK(hipMemcpy(d_chars, h.data(), h.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(d_lens, lens.data(), 4 * B, hipMemcpyHostToDevice));
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<char *> va(NS); std::vector<std::vector<hipMemGenericAllocationHandle_t>> pa(NS);
    for (size_t s = 0; s < NS; ++s) {
        CK(hipMemAddressReserve((void **)&va[s], tot, chunk, nullptr, 0));
        pa[s].resize(tot / chunk);
        for (auto &hd : pa[s]) CK(hipMemCreate(&hd, chunk, &prop, 0));
        void *junk; CK(hipMalloc(&junk, (size_t)(100 + 300 * s) << 20));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int round = 0; round < 2; ++round) {
        printf("round %d (rows: virtual range, columns: physical set), us per launch\n", round);
        for (size_t v = 0; v < NS; ++v) {
            printf("  va %p:", (void *)va[v]);
            for (size_t p = 0; p < NS; ++p) {
                for (size_t c = 0; c < pa[p].size(); ++c) CK(hipMemMap(va[v] + c * chunk, chunk, 0, pa[p][c], 0));
                CK(hipMemSetAccess(va[v], tot, &acc, 1));
                uint32_t *rec = (uint32_t *)va[v]; uint16_t *msk = (uint16_t *)(va[v] + rec_bytes);
                for (int i = 0; i < 3; ++i) HK(hrx_witness_batch_device_layout(ctx, HRX_LAYOUT_POSITION_MAJOR, d_chars, M, d_lens, B, M, rec, msk, d_st, nullptr));
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < 20; ++i) HK(hrx_witness_batch_device_layout(ctx, HRX_LAYOUT_POSITION_MAJOR, d_chars, M, d_lens, B, M, rec, msk, d_st, nullptr));
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf(" %7.1f", ms * 1e3 / 20);
                CK(hipMemUnmap(va[v], tot));
            }
            printf("\n");
        }
    }
    return 0;
}
