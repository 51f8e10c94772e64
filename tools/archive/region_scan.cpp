// region_scan — the device memory taken 1 GiB at a time (allocation order: the driver hands out VRAM top-down), every GiB
// classified against a few reference GiBs by the bandwidth of two concurrent write streams (6.3 TB/s within a region of the
// physical address space, 7.5 TB/s across two: tools/halves_probe.cpp).  (tools only; gfx950)
// build: hipcc --offload-arch=gfx950 -O2 -o region_scan region_scan.cpp ; usage: region_scan [GiB to take = 270] [chunk MiB = 1024]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void fill2(unsigned char *a, unsigned char *b, size_t bytes) {   // 512 waves on each region
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    unsigned char *p = ((wave & 1u) ? b : a) + ((size_t)(wave >> 1) << 10) + lane * 16;
    const v4u32 v = {1, 2, 3, 4};
    for (size_t off = 0; off + (512u << 10) <= bytes; off += 512u << 10) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p + off), "v"(v) : "memory");
}
int main(int argc, char **argv) {
    const size_t take = argc > 1 ? atol(argv[1]) : 270, chunk = (size_t)(argc > 2 ? atol(argv[2]) : 1024) << 20;
    size_t fr, tot; CK(hipMemGetInfo(&fr, &tot));
    printf("free %.1f of %.1f GiB\n", fr / 1073741824.0, tot / 1073741824.0);
    std::vector<unsigned char *> c;
    for (size_t i = 0; i < take * ((size_t)1 << 30) / chunk; ++i) {
        unsigned char *p = nullptr;
        if (hipExtMallocWithFlags((void **)&p, chunk, hipDeviceMallocContiguous) != hipSuccess) { (void)hipGetLastError(); break; }
        c.push_back(p);
    }
    printf("%zu chunks of %zu MiB\n", c.size(), chunk >> 20);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto bw = [&](unsigned char *a, unsigned char *b) {
        double best = 1e30;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(fill2, dim3(256), dim3(256), 0, 0, a, b, chunk); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r) best = std::min(best, (double)ms);
        }
        return 2.0 * chunk / best / 1e6;
    };
    for (size_t ref : {(size_t)0, c.size() / 3, c.size() / 2, 2 * c.size() / 3, c.size() - 1}) {
        printf("against chunk %zu (TB/s x 10):", ref);
        for (size_t i = 0; i < c.size(); ++i) printf(" %d", i == ref ? 0 : (int)(bw(c[ref], c[i]) / 100.0 + 0.5));
        printf("\n");
    }
    return 0;
}
