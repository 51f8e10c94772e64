// wpattern.cpp — on-box probe (profiling only): HBM write bandwidth of the witness store pattern as a function of the
// contiguous run each string receives per visit.  B "strings" of RB bytes each (string-major); 1024 waves; a wave owns 64
// strings and visits them run by run: every wave-instruction writes 1 KiB = (1024/S) strings x S contiguous bytes.
//   hipcc --offload-arch=gfx950 -O3 tools/wpattern.cpp -o tools/wpattern && tools/wpattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int S, bool NT, bool TM>
__global__ __launch_bounds__(256) void wk(uint4 *out, size_t RB, int nstr, int delay, int spw) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    constexpr int LPS = S / 16;        // lanes per string run
    constexpr int SPI = 64 / LPS;      // strings per instruction
    const int s0 = lane / LPS, c = lane % LPS;
    for (size_t off = 0; off < RB; off += S) {       // one "tile" = S bytes for each of the wave's 64 strings
        for (int it = 0; it < spw / SPI; ++it) {
            const size_t str = (size_t)wave * spw + it * SPI + s0;
            uint4 *p = TM ? (uint4 *)((char *)out + (off / S) * ((size_t)nstr * S) + str * S) + c   // tile-major: [RB/S][nstr][S]
                          : (uint4 *)((char *)out + str * RB + off) + c;                          // string-major: [nstr][RB]
            const uint4 v = make_uint4(lane, it, (unsigned)off, wave);
            typedef unsigned v4 __attribute__((ext_vector_type(4)));
            if (NT) __builtin_nontemporal_store(v4{v.x, v.y, v.z, v.w}, (v4 *)p); else *p = v;
        }
        for (int i = 0; i < delay; ++i) __builtin_amdgcn_s_sleep(8);   // stand-in for the walk between store bursts
    }
}

__global__ void fillk(uint4 *out, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) out[i] = make_uint4(1, 2, 3, (unsigned)i);
}

template <int S, bool NT, bool TM = false>
static void run(uint4 *d, size_t RB, int nstr, int delay, int spw = 64) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int waves = nstr / spw;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((wk<S, NT, TM>), dim3(waves / 4), dim3(256), 0, 0, d, RB, nstr, delay, spw);
    CK(hipEventRecord(e0, 0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((wk<S, NT, TM>), dim3(waves / 4), dim3(256), 0, 0, d, RB, nstr, delay, spw);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, bytes = (double)nstr * RB;
    printf("run spw=%d S=%4d B  nt=%d tilemajor=%d delay=%d: %8.1f us  %.2f TB/s\n", spw, S, (int)NT, (int)TM, delay, us, bytes / (us * 1e-6) / 1e12);
}

int main(int argc, char **argv) {
    const int nstr = 65536; const size_t RB = argc > 1 ? atol(argv[1]) : 6144;  // 6 KiB per string ~ records+masked of 1024 rows
    uint4 *d; CK(hipMalloc(&d, (size_t)nstr * RB));
    for (int spw : {64, 32, 16, 8}) { run<128, false>(d, RB, nstr, 0, spw); run<256, false>(d, RB, nstr, 0, spw); run<128, false, true>(d, RB, nstr, 0, spw); }
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int g : {1024, 4096, 16384}) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(fillk, dim3(g), dim3(256), 0, 0, d, (size_t)nstr * RB / 16);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("grid-stride fill, %d blocks: %.1f us %.2f TB/s\n", g, ms * 50, (double)nstr * RB / (ms * 50e-6) / 1e12);
        }
    }
    for (int delay : {0}) {
        run<64, false>(d, RB, nstr, delay); run<128, false>(d, RB, nstr, delay); run<256, false>(d, RB, nstr, delay);
        run<1024, false>(d, RB, nstr, delay);
        run<64, false, true>(d, RB, nstr, delay); run<128, false, true>(d, RB, nstr, delay); run<256, false, true>(d, RB, nstr, delay);
        run<1024, false, true>(d, RB, nstr, delay);
    }
    return 0;
}
