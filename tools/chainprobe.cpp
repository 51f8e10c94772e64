// chainprobe.cpp — on-box probe: cycles per step of a dependent LDS pointer chase, the walk's critical path.
//   one workgroup per CU, W waves per workgroup (1 wave per SIMD at W = 4), every lane chases its own chain through a
//   table in LDS; variants differ in the read width and in the VALU op between two reads:
//     0: ds_read_b32, v_and_or_b32            (witness_pm_kernel's chain)
//     1: ds_read_b64, v_mad_u32_u16 (x8)      (witness_pp_kernel's chain)
//     2: ds_read_b64, v_and + v_lshl_add
//     3: ds_read_b64, v_mad_u32_u24
//     4: ds_read_b64, v_add_u32 (entry holds the byte address)
//     5: ds_read_b32, v_add_u32
//     6: ds_read_b64, v_add_u32, plus 12 independent VALU ops per step (post-processing stand-in)
//     7: ds_read_b32, v_and_or, plus 12 independent VALU ops per step
//   Reported: s_memtime ticks (100 MHz) and core cycles (from the wall time of the launch and the steps) per step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kEntries = 8192;   // 64 KiB of 8-byte entries / 32 KiB of 4-byte entries
constexpr int kSteps = 4096;

template <int V>
__global__ __launch_bounds__(512) void chase(const unsigned *tab, unsigned *out, unsigned long long *ticks, int spread) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned *t32 = reinterpret_cast<unsigned *>(smem);
    for (int i = threadIdx.x; i < kEntries * 2; i += blockDim.x) t32[i] = tab[i];
    __syncthreads();
    const unsigned lane = threadIdx.x & 63;
    unsigned idx = spread ? (lane * 8u) : 0u;     // per-lane pair index (bytes), constant: only the table value changes the address
    unsigned lo = (threadIdx.x * 37u) % kEntries; // start entry
    unsigned acc = 0, x0 = lane, x1 = lane * 3, x2 = lane * 5, x3 = lane * 7;
    typedef __attribute__((address_space(3))) const unsigned lds_u32_t;
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) const v2u lds_v2_t;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 8
    for (int s = 0; s < kSteps; ++s) {
        unsigned addr;
        if (V == 0 || V == 7) { asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(addr) : "v"(lo), "v"(0xfffcu), "v"(0u)); }
        else if (V == 1) { asm volatile("v_mad_u32_u16 %0, %1, 8, %2" : "=v"(addr) : "v"(lo), "v"(0u)); }
        else if (V == 2) { unsigned t; asm volatile("v_and_b32 %0, 0xffff, %1" : "=v"(t) : "v"(lo)); asm volatile("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(addr) : "v"(t), "v"(0u)); }
        else if (V == 3) { asm volatile("v_mad_u32_u24 %0, %1, 8, %2" : "=v"(addr) : "v"(lo), "v"(0u)); }
        else { asm volatile("v_add_u32 %0, %1, %2" : "=v"(addr) : "v"(lo), "v"(0u)); }
        if (V == 0 || V == 5 || V == 7) {
            lo = *(lds_u32_t *)(unsigned long)addr;
        } else {
            const v2u r = *(lds_v2_t *)(unsigned long)addr;
            lo = r.x; acc ^= r.y;
        }
        if (V == 6 || V == 7) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x0) : "v"(x1), "v"(0x06020c04u));
                asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x1) : "v"(x2), "v"(x3));
                asm volatile("v_alignbit_b32 %0, %0, %1, 3" : "+v"(x2) : "v"(x3));
                asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(x3) : "v"(x0), "v"(x1));
            }
        }
        (void)idx;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = lo ^ acc ^ x0 ^ x1 ^ x2 ^ x3;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int V> static void run(const char *name, const unsigned *d_tab, unsigned *d_out, unsigned long long *d_ticks, int waves) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(chase<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipLaunchKernelGGL(chase<V>, dim3(256), dim3(64 * waves), 65536, 0, d_tab, d_out, d_ticks, 1);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(chase<V>, dim3(256), dim3(64 * waves), 65536, 0, d_tab, d_out, d_ticks, 1);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long t; CK(hipMemcpy(&t, d_ticks, 8, hipMemcpyDeviceToHost));
    printf("%-44s waves/CU %d: %.1f ns/step (launch %.1f us), s_memtime %.2f ticks/step\n", name, waves, ms * 1e6 / kSteps, ms * 1e3, (double)t / kSteps);
}

int main() {
    // table: entry i -> a pseudo-random next entry; 8-byte view: lo = next (address / 8 for the x8 variants, byte address for 4/6), hi = junk
    std::vector<unsigned> t8(kEntries * 2), t8b(kEntries * 2), t4(kEntries * 2);
    unsigned s = 12345;
    for (int i = 0; i < kEntries; ++i) {
        s = s * 1664525u + 1013904223u;
        const unsigned nxt = (s >> 8) % kEntries;
        t8[2 * i] = nxt | 0xab000000u; t8[2 * i + 1] = s;                 // address / 8 in the low 16 bits, payload above
        t8b[2 * i] = nxt * 8; t8b[2 * i + 1] = s;                         // byte address
    }
    for (int i = 0; i < kEntries * 2; ++i) { s = s * 1664525u + 1013904223u; t4[i] = ((s >> 8) % kEntries) * 4; }
    unsigned *d8, *d8b, *d4, *d_out; unsigned long long *d_ticks;
    CK(hipMalloc(&d8, kEntries * 8)); CK(hipMalloc(&d8b, kEntries * 8)); CK(hipMalloc(&d4, kEntries * 8)); CK(hipMalloc(&d_out, 256 * 512 * 4)); CK(hipMalloc(&d_ticks, 256 * 8));
    CK(hipMemcpy(d8, t8.data(), kEntries * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d8b, t8b.data(), kEntries * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d4, t4.data(), kEntries * 8, hipMemcpyHostToDevice));
    for (int waves : {4, 8}) {
        run<0>("0: ds_read_b32 + v_and_or", d4, d_out, d_ticks, waves);
        run<5>("5: ds_read_b32 + v_add", d4, d_out, d_ticks, waves);
        run<1>("1: ds_read_b64 + v_mad_u32_u16", d8, d_out, d_ticks, waves);
        run<2>("2: ds_read_b64 + v_and + v_lshl_add", d8, d_out, d_ticks, waves);
        run<3>("3: ds_read_b64 + v_mad_u32_u24", d8, d_out, d_ticks, waves);
        run<4>("4: ds_read_b64 + v_add", d8b, d_out, d_ticks, waves);
        run<6>("6: ds_read_b64 + v_add + 12 VALU", d8b, d_out, d_ticks, waves);
        run<7>("7: ds_read_b32 + v_and_or + 12 VALU", d4, d_out, d_ticks, waves);
    }
    return 0;
}
