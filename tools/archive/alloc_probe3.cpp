// alloc_probe3 — which ADDRESS BITS should the 1024 concurrently writing waves differ in?  (tools only; gfx950)
// A 4-GiB buffer = 2^22 pieces of 1 KiB (address bits 10..31).  The wave index (10 bits) is deposited into a chosen set of ten
// of those bit positions, the time step (12 bits) into the remaining twelve, low to high; every piece is written exactly once.
// build: hipcc --offload-arch=gfx950 -O2 -o alloc_probe3 alloc_probe3.cpp ; usage: alloc_probe3 [spec ...], spec = e.g. 10-15,20-23
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
__device__ __host__ inline uint32_t deposit(uint32_t v, uint32_t mask) {
    uint32_t r = 0;
    for (uint32_t b = 0; mask; mask &= mask - 1, ++b) if (v >> b & 1u) r |= mask & -mask;
    return r;
}
__global__ __launch_bounds__(256) void fill_bits(unsigned char *p, uint32_t maskW, uint32_t maskQ, uint32_t steps, int nt) {
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const v4u32 v = {1, 2, 3, 4};
    unsigned char *a = p + deposit(wave, maskW) + lane * 16;
    uint32_t q = 0;
    for (uint32_t s = 0; s < steps; ++s) {
        if (nt) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(a + q), "v"(v) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(a + q), "v"(v) : "memory");
        q = ((q | ~maskQ) + 1u) & maskQ;   // the next value of the time counter, in place in its bit positions
    }
}
int main(int argc, char **argv) {
    unsigned char *p; CK(hipMalloc((void **)&p, (size_t)4 << 30));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<std::string> specs;
    for (int i = 1; i < argc; ++i) specs.push_back(argv[i]);
    for (auto &sp : specs) {
        uint32_t maskW = 0;
        char buf[256]; strncpy(buf, sp.c_str(), 255); buf[255] = 0;
        for (char *t = strtok(buf, ","); t; t = strtok(nullptr, ",")) {
            int lo = atoi(t), hi = strchr(t, '-') ? atoi(strchr(t, '-') + 1) : lo;
            for (int b = lo; b <= hi; ++b) maskW |= 1u << b;
        }
        if (__builtin_popcount(maskW) != 10 || (maskW & 0x3ffu)) { printf("%-28s bad spec\n", sp.c_str()); continue; }
        const uint32_t maskQ = ~maskW & 0xfffffc00u;
        double t[2];
        for (int nt = 0; nt < 2; ++nt) {
            double best = 1e30;
            for (int r = 0; r < 4; ++r) {
                CK(hipEventRecord(e0)); hipLaunchKernelGGL(fill_bits, dim3(256), dim3(256), 0, 0, p, maskW, maskQ, 4096u, nt); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r) best = std::min(best, (double)ms);
            }
            t[nt] = 4294.967296 / best;
        }
        printf("%-28s wb %5.0f  nt %5.0f GB/s\n", sp.c_str(), t[0], t[1]);
    }
    return 0;
}
