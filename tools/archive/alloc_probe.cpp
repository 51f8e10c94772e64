// alloc_probe — does the write bandwidth of a buffer depend on the ALLOCATION it lives in?  (tools only; gfx950)
// tools/set_probe*.py: the cfg 3 launch takes 0.97 .. 1.19 ms depending on which allocation holds its outputs, whatever the
// offsets inside it.  This probe writes buffers obtained in different ways with two patterns and prints GB/s per buffer:
//   linear : every wave writes consecutive 1-KiB pieces, grid-stride (what a plain fill does)
//   column : 1024 waves, wave w writes the 1-KiB piece w of every 1-MiB slab, top to bottom (the position-major walkers)
// build: hipcc --offload-arch=gfx950 -O2 -o alloc_probe alloc_probe.cpp
// usage: alloc_probe [GiB per buffer = 4] [buffers = 6] [mode: malloc | vmm:<chunk MiB> | vmmshuf:<chunk MiB> | vmmmul:<chunk MiB>:<odd multiplier> | vmmrot:<chunk MiB>:<bits>]
//   (vmm*: the buffer's physical chunks mapped in order / shuffled / chunk c -> handle (c * m) mod n / chunk index rotated)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_nt(void *p, v4u32 v) { asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(v) : "memory"); }

__global__ __launch_bounds__(256) void fill_linear(unsigned char *p, size_t bytes, int nt) {
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (size_t)gridDim.x * 4, lane = threadIdx.x & 63;
    const v4u32 v = {1, 2, 3, 4};
    for (size_t k = wave; k < bytes >> 10; k += nw) {
        void *q = p + (k << 10) + lane * 16;
        if (nt) st_nt(q, v); else *(v4u32 *)q = v;
    }
}
__global__ __launch_bounds__(256) void fill_column(unsigned char *p, size_t bytes, int nt) {   // grid 256 x 4 waves = 1024 columns
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const v4u32 v = {1, 2, 3, 4};
    for (size_t q = 0; q < bytes >> 20; ++q) {
        void *a = p + (q << 20) + (wave << 10) + lane * 16;
        if (nt) st_nt(a, v); else *(v4u32 *)a = v;
    }
}
static double time_kernel(void (*k)(unsigned char *, size_t, int), int grid, unsigned char *p, size_t bytes, int nt) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, p, bytes, nt);
    double best = 1e30;
    for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, p, bytes, nt); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, (double)ms);
    }
    return bytes / best / 1e6;   // GB/s
}
int main(int argc, char **argv) {
    const size_t gib = argc > 1 ? atol(argv[1]) : 4; const int nbuf = argc > 2 ? atoi(argv[2]) : 6;
    const char *mode = argc > 3 ? argv[3] : "malloc";
    const size_t bytes = gib << 30;
    std::vector<unsigned char *> bufs;
    for (int i = 0; i < nbuf; ++i) {
        unsigned char *p = nullptr;
        if (!strcmp(mode, "malloc")) { CK(hipMalloc((void **)&p, bytes)); }
        else {
            const bool shuf = !strncmp(mode, "vmmshuf:", 8);
            const size_t chunk = (size_t)atol(strchr(mode, ':') + 1) << 20;
            hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
            size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
            if (i == 0) printf("vmm granularity %zu KiB, chunk %zu MiB\n", gran >> 10, chunk >> 20);
            CK(hipMemAddressReserve((void **)&p, bytes, (size_t)1 << 30, nullptr, 0));
            const size_t n = bytes / chunk;
            std::vector<hipMemGenericAllocationHandle_t> hs(n);
            for (size_t c = 0; c < n; ++c) CK(hipMemCreate(&hs[c], chunk, &prop, 0));
            std::vector<size_t> order(n); for (size_t c = 0; c < n; ++c) order[c] = c;
            if (shuf) { std::mt19937 g(1234 + i); std::shuffle(order.begin(), order.end(), g); }
            if (!strncmp(mode, "vmmmul:", 7)) { const size_t m = (size_t)atol(strrchr(mode, ':') + 1); for (size_t c = 0; c < n; ++c) order[c] = (c * m) % n; }   // m odd, n a power of two
            if (!strncmp(mode, "vmmrot:", 7)) {   // rotate the chunk index left by r bits
                const int r = atoi(strrchr(mode, ':') + 1); int nb = 0; while (((size_t)1 << nb) < n) ++nb;
                for (size_t c = 0; c < n; ++c) order[c] = ((c << r) | (c >> (nb - r))) & (n - 1);
            }
            for (size_t c = 0; c < n; ++c) CK(hipMemMap(p + c * chunk, chunk, 0, hs[order[c]], 0));
            hipMemAccessDesc d = {}; d.location = prop.location; d.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(p, bytes, &d, 1));
        }
        bufs.push_back(p);
        void *junk; CK(hipMalloc(&junk, (size_t)(37 + 64 * i) << 20));   // shift what comes next
    }
    for (int round = 0; round < 2; ++round)
        for (int i = 0; i < nbuf; ++i)
            printf("%s buf %d %p: linear wb %6.0f nt %6.0f | column wb %6.0f nt %6.0f GB/s\n", mode, i, (void *)bufs[i],
                   time_kernel(fill_linear, 2048, bufs[i], bytes, 0), time_kernel(fill_linear, 2048, bufs[i], bytes, 1),
                   time_kernel(fill_column, 256, bufs[i], bytes, 0), time_kernel(fill_column, 256, bufs[i], bytes, 1));
    return 0;
}
