// mixprobe.cpp — on-box probe: the position-major traffic mix of the witness kernel without any compute.
// 1024 waves (256 x 256 threads); wave w owns strings 64w..64w+63.  Per quad of rows q: 1 KiB of records into slab q;
// per octet: 1 KiB of masked rows; per 16 rows: 1 KiB of input read (sum kept live).  Variants drop streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ __launch_bounds__(256) void mix(const uint4 *in, uint4 *rec, uint4 *msk, unsigned *sink, int B, int streams, int pause) {
    const size_t gt = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // = string index
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int q = 0; q < 256; ++q) {
        if (streams & 1) rec[(size_t)q * B + gt] = make_uint4(q, 1, 2, 3);
        if ((streams & 2) && (q & 1)) msk[(size_t)(q >> 1) * B + gt] = make_uint4(0, 0, 0, q);
        if ((streams & 4) && (q & 3) == 0) { const uint4 v = in[(size_t)(q >> 2) * B + gt]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
        for (int k = 0; k < pause; ++k) __builtin_amdgcn_s_sleep(2);
    }
    if (acc.x == 0x12345678u) sink[0] = acc.y ^ acc.z ^ acc.w;
}
int main() {
    const int B = 65536;
    uint4 *in, *rec, *msk; unsigned *sink;
    CK(hipMalloc(&in, (size_t)B * 1024)); CK(hipMalloc(&rec, (size_t)B * 4096)); CK(hipMalloc(&msk, (size_t)B * 2048)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(in, 1, (size_t)B * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int pause : {0, 2}) for (int streams : {1, 2, 4, 3, 5, 7}) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(mix, dim3(256), dim3(256), 0, 0, in, rec, msk, sink, B, streams, pause);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(mix, dim3(256), dim3(256), 0, 0, in, rec, msk, sink, B, streams, pause);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)B * ((streams & 1 ? 4096 : 0) + (streams & 2 ? 2048 : 0) + (streams & 4 ? 1024 : 0));
        printf("pause %d streams rec=%d msk=%d in=%d: %.1f us  %.2f TB/s\n", pause, streams & 1, (streams >> 1) & 1, (streams >> 2) & 1, ms * 50, bytes / (ms / 20 * 1e-3) / 1e12);
    }
    return 0;
}
