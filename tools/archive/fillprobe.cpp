// fillprobe.cpp — on-box probe: what limits a plain sequential fill?  waves per CU x bytes per thread.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int PER>
__global__ void fillk(uint4 *out, size_t n16) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
#pragma unroll
    for (int k = 0; k < PER; ++k) { if (i < n16) out[i] = make_uint4(1, 2, 3, (unsigned)i); i += stride; }
}
__global__ void fill_loop(uint4 *out, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) out[i] = make_uint4(1, 2, 3, (unsigned)i);
}
// every wave writes its own contiguous chunk, 1 KiB per instruction, optionally pausing between stores like a walker does
__global__ void fill_blocked(uint4 *out, size_t n16, int pause) {
    const size_t waves = (size_t)gridDim.x * blockDim.x / 64, w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64;
    const size_t per = n16 / waves;  // uint4 per wave
    uint4 *p = out + w * per + (threadIdx.x & 63);
    for (size_t i = 0; i < per; i += 64) { p[i] = make_uint4(1, 2, 3, (unsigned)i); for (int k = 0; k < pause; ++k) __builtin_amdgcn_s_sleep(2); }
}
__global__ void fill_strided(uint4 *out, size_t n16, int pause) {  // grid-stride: all waves inside one moving 1-MiB window
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { out[i] = make_uint4(1, 2, 3, (unsigned)i); for (int k = 0; k < pause; ++k) __builtin_amdgcn_s_sleep(2); }
}
int main() {
    const size_t bytes = (size_t)448 << 20, n16 = bytes / 16;
    uint4 *d; CK(hipMalloc(&d, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 20; ++i) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-40s %.1f us  %.2f TB/s\n", name, ms * 50, bytes / (ms / 20 * 1e-3) / 1e12);
    };
    time("1 x 16B per thread, 256 thr", [&] { hipLaunchKernelGGL(fillk<1>, dim3((n16 + 255) / 256), dim3(256), 0, 0, d, n16); });
    time("4 x 16B per thread, 256 thr", [&] { hipLaunchKernelGGL(fillk<4>, dim3((n16 / 4 + 255) / 256), dim3(256), 0, 0, d, n16); });
    time("16 x 16B per thread, 256 thr", [&] { hipLaunchKernelGGL(fillk<16>, dim3((n16 / 16 + 255) / 256), dim3(256), 0, 0, d, n16); });
    time("64 x 16B per thread, 256 thr", [&] { hipLaunchKernelGGL(fillk<64>, dim3((n16 / 64 + 255) / 256), dim3(256), 0, 0, d, n16); });
    for (int g : {256, 512, 1024, 2048, 4096, 8192}) {
        char nm[64]; snprintf(nm, 64, "loop, %d blocks x 256", g);
        time(nm, [&] { hipLaunchKernelGGL(fill_loop, dim3(g), dim3(256), 0, 0, d, n16); });
    }
    for (int g : {256, 512, 1024}) {
        char nm[64]; snprintf(nm, 64, "loop, %d blocks x 1024", g);
        time(nm, [&] { hipLaunchKernelGGL(fill_loop, dim3(g), dim3(1024), 0, 0, d, n16); });
    }
    for (int pause : {0, 1, 2}) {
        char nm[64];
        snprintf(nm, 64, "blocked, 256x256, pause %d", pause);
        time(nm, [&] { hipLaunchKernelGGL(fill_blocked, dim3(256), dim3(256), 0, 0, d, n16, pause); });
        snprintf(nm, 64, "strided, 256x256, pause %d", pause);
        time(nm, [&] { hipLaunchKernelGGL(fill_strided, dim3(256), dim3(256), 0, 0, d, n16, pause); });
    }
    return 0;
}
