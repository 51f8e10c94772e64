// region_map2d — two concurrent write streams at every pair of places of the largest contiguous allocation the device gives:
// which places of the physical address space belong together?  (tools only; gfx950)
// Prints, per place a (rows, every `step` GiB), the combined GB/s / 100 of streams at a and b; 63 = same region, 74 = different
// (tools/halves_probe.cpp).  build: hipcc --offload-arch=gfx950 -O2 -o region_map2d region_map2d.cpp
// usage: region_map2d [slab GiB = 256] [step GiB = 8] [bytes per stream MiB = 256]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void fill2(unsigned char *a, unsigned char *b, size_t bytes) {   // 512 waves on each place
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    unsigned char *p = ((wave & 1u) ? b : a) + ((size_t)(wave >> 1) << 10) + lane * 16;
    const v4u32 v = {1, 2, 3, 4};
    for (size_t off = 0; off + (512u << 10) <= bytes; off += 512u << 10) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p + off), "v"(v) : "memory");
}
int main(int argc, char **argv) {
    size_t slab = argc > 1 ? atol(argv[1]) : 256; const size_t step = argc > 2 ? atol(argv[2]) : 8, bytes = (size_t)(argc > 3 ? atol(argv[3]) : 256) << 20;
    size_t fr, tot; CK(hipMemGetInfo(&fr, &tot));
    unsigned char *p = nullptr;
    while (slab >= 32 && hipExtMallocWithFlags((void **)&p, slab << 30, hipDeviceMallocContiguous) != hipSuccess) { (void)hipGetLastError(); slab -= 16; }
    printf("free %.1f of %.1f GiB; contiguous slab of %zu GiB at %p\n", fr / 1073741824.0, tot / 1073741824.0, slab, (void *)p);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto bw = [&](size_t a, size_t b) {
        double best = 1e30;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(fill2, dim3(256), dim3(256), 0, 0, p + (a << 30), p + (b << 30), bytes); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r) best = std::min(best, (double)ms);
        }
        return 2.0 * bytes / best / 1e6;
    };
    printf("      "); for (size_t b = 0; b + 1 <= slab; b += step) printf("%4zu", b); printf("\n");
    for (size_t a = 0; a + 1 <= slab; a += step) {
        printf("%4zu: ", a);
        for (size_t b = 0; b + 1 <= slab; b += step) printf("%4d", a == b ? 0 : (int)(bw(a, b) / 100.0 + 0.5));
        printf("\n");
    }
    return 0;
}
