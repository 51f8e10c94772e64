// alloc_probe2 — which address distances between concurrently written streams cost write bandwidth?  (tools only; gfx950)
// One hipMalloc'd buffer.  1024 waves in K streams of 1024/K waves; stream k writes, top to bottom, consecutive windows of
// (1024/K) KiB starting at k * delta (wave w of the stream: the 1-KiB piece w of each window).  K = 1 is alloc_probe's
// "column" pattern.  build: hipcc --offload-arch=gfx950 -O2 -o alloc_probe2 alloc_probe2.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void fill_streams(unsigned char *p, size_t per_stream, size_t delta, uint32_t K) {
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t wps = 1024u / K, k = wave / wps, w = wave % wps;
    const size_t win = (size_t)wps << 10;
    const v4u32 v = {1, 2, 3, 4};
    unsigned char *a = p + (size_t)k * delta + ((size_t)w << 10) + lane * 16;
    for (size_t q = 0; q < per_stream / win; ++q) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(a + q * win), "v"(v) : "memory");
}
int main(int argc, char **argv) {
    const size_t total = (size_t)(argc > 1 ? atol(argv[1]) : 24) << 30;
    unsigned char *p; CK(hipMalloc((void **)&p, total));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (uint32_t K : {1u, 2u, 4u, 8u, 16u}) {
        const size_t per_stream = ((size_t)4 << 30) / K;
        printf("K=%2u (window %4u KiB):", K, 1024u / K);
        for (size_t dm : {1, 2, 3, 4, 6, 8, 16, 17, 32, 64, 65, 128, 256, 512, 1024}) {
            const size_t delta = dm << 20;
            if (K == 1 && dm > 1) break;
            if ((K - 1) * delta + per_stream > total) { printf(" %zuM:  -  ", dm); continue; }
            double best = 1e30;
            for (int r = 0; r < 4; ++r) {
                CK(hipEventRecord(e0)); hipLaunchKernelGGL(fill_streams, dim3(256), dim3(256), 0, 0, p, per_stream, delta, K); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r) best = std::min(best, (double)ms);
            }
            printf(" %zuM:%5.0f", dm, per_stream * K / best / 1e6);
        }
        printf("\n");
    }
    return 0;
}
