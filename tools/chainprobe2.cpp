// chainprobe2.cpp — round 4, cfg 5: what a dependent LDS byte-table chase costs per step while the CU does what the witness kernel does around it.
//   One workgroup of 16 waves per CU.  Roles by wave index:
//     chain waves (NA)   every lane chases next = T[state << 8 | byte] through a 64-KiB random next-state table in LDS (v_perm_b32 + ds_read_u8: the BYTE walker's chain)
//     LDS waves   (NB)   independent random ds_read_b32 + ds_write_b128 / ds_read_b128 traffic (the pair-tag lookups and hand-over blocks of the other waves)
//     VALU waves  (ND, the LAST waves of the workgroup) pure vector arithmetic: what the other waves' bit work costs the chain at the SIMD's issue arbiter; prio: the chain waves raise their priority
//     store waves (NC)   stream 16-byte nt stores over a large buffer (the launch's record / masked-row traffic: what the memory system and the clocks see)
//   Reported per variant: ns per chain step (wall clock of the launch / steps), shader cycles per step (s_memtime) and the resulting clock.
//   build: hipcc --offload-arch=gfx950 -O3 -o chainprobe2 chainprobe2.cpp      run: ./chainprobe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kSteps = 8192;
typedef __attribute__((address_space(3))) const unsigned char lds_u8_t;
typedef __attribute__((address_space(3))) const unsigned lds_u32_t;
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4u lds_v4_t;

__global__ __launch_bounds__(1024) void probe(const unsigned char *tab, unsigned *out, unsigned long long *ticks, unsigned char *sink, size_t sink_bytes,
                                              int NA, int NB, int NC, volatile int *stop, int ND, int prio) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 65536 / 16; i += blockDim.x) reinterpret_cast<uint4 *>(smem)[i] = reinterpret_cast<const uint4 *>(tab)[i];
    __shared__ int done;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((int)wave < NA) {
        if (prio) __builtin_amdgcn_s_setprio(3);
        unsigned state = (threadIdx.x * 37u) & 255u;
        unsigned bytes = threadIdx.x * 2654435761u;
        const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 8
        for (int s = 0; s < kSteps; ++s) {
            bytes = bytes * 1664525u + 1013904223u;                       // (off the chain: the input bytes are there before the state is)
            const unsigned addr = __builtin_amdgcn_perm(state, bytes >> 24, 0x0c0c0400u);
            asm volatile("" ::: "memory");
            state = *(lds_u8_t *)(unsigned long)addr;
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        out[blockIdx.x * 1024 + threadIdx.x] = state;
        if (lane == 0) ticks[blockIdx.x * 16 + wave] = t1 - t0;
        if (lane == 0) atomicAdd(&done, 1);
    } else if ((int)wave < NA + NB) {
        unsigned x = threadIdx.x * 747796405u + 1u, acc = 0;
        const unsigned blk = 65536u + (wave - NA) * 4096u + lane * 16u;   // this wave's 4-KiB hand-over block behind the table
        while (*(volatile int *)&done < NA) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                x = x * 1664525u + 1013904223u;
                acc ^= *(lds_u32_t *)(unsigned long)((x >> 14) & 0xfffcu);   // random dword of the table
            }
            *(lds_v4_t *)(unsigned long)blk = v4u{acc, x, acc, x};
            const v4u r = *(lds_v4_t *)(unsigned long)(blk ^ 1024u);
            acc ^= r.x;
        }
        out[blockIdx.x * 1024 + threadIdx.x] = acc;
    } else if ((int)wave >= 16 - ND) {
        // VALU waves: dependent-free vector arithmetic, no memory at all (what a finisher / recorder wave's bit work looks like to the SIMD's issue arbiter)
        unsigned x0 = threadIdx.x, x1 = lane * 3u, x2 = lane * 5u, x3 = lane * 7u;
        while (*(volatile int *)&done < NA) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x0) : "v"(x1), "v"(0x06020c04u));
                asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x1) : "v"(x2), "v"(x3));
                asm volatile("v_alignbit_b32 %0, %0, %1, 3" : "+v"(x2) : "v"(x3));
                asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(x3) : "v"(x0), "v"(x1));
            }
        }
        out[blockIdx.x * 1024 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3;
    } else if ((int)wave < NA + NB + NC) {
        const size_t per_wave = 1024;                                       // one store instruction = 1 KiB contiguous
        size_t pos = ((size_t)blockIdx.x * 16 + wave) * per_wave;
        const size_t stride = (size_t)gridDim.x * 16 * per_wave;
        const v4u v = v4u{1u, 2u, 3u, 4u};
        while (*(volatile int *)&done < NA) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(sink + (pos % sink_bytes) + lane * 16), "v"(v) : "memory");
                pos += stride;
            }
        }
    }
}

int main() {
    std::vector<unsigned char> t(65536);
    unsigned s = 12345;
    for (auto &b : t) { s = s * 1664525u + 1013904223u; b = (unsigned char)(s >> 24); }
    unsigned char *d_tab, *d_sink; unsigned *d_out; unsigned long long *d_ticks; int *d_stop;
    const size_t sink_bytes = (size_t)8 << 30;
    CK(hipMalloc(&d_tab, 65536)); CK(hipMalloc(&d_out, 256 * 1024 * 4)); CK(hipMalloc(&d_ticks, 256 * 16 * 8)); CK(hipMalloc(&d_sink, sink_bytes)); CK(hipMalloc(&d_stop, 4));
    CK(hipMemcpy(d_tab, t.data(), 65536, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 16 * 4096));
    struct V { const char *name; int na, nb, nc, nd, prio; } vs[] = {
        {"4 chain waves alone                              ", 4, 0, 0},
        {"4 chain + 4 LDS-traffic waves                    ", 4, 4, 0},
        {"4 chain + 8 LDS-traffic waves                    ", 4, 8, 0},
        {"4 chain + 8 store waves (memory system loaded)   ", 4, 0, 8},
        {"4 chain + 4 LDS-traffic + 8 store waves          ", 4, 4, 8},
        {"8 chain waves alone                              ", 8, 0, 0},
        {"8 chain + 8 store waves                          ", 8, 0, 8},
        {"4 chain + 4 VALU waves (2 waves per SIMD)        ", 4, 0, 0, 4, 0},
        {"4 chain + 8 VALU waves (3 per SIMD)              ", 4, 0, 0, 8, 0},
        {"4 chain + 12 VALU waves (4 per SIMD)             ", 4, 0, 0, 12, 0},
        {"4 chain (s_setprio 3) + 12 VALU waves            ", 4, 0, 0, 12, 1},
        {"4 chain + 4 LDS + 8 VALU waves                   ", 4, 4, 0, 8, 0},
        {"4 chain (s_setprio 3) + 4 LDS + 8 VALU waves     ", 4, 4, 0, 8, 1},
    };
    for (const V &v : vs) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(probe, dim3(256), dim3(1024), 65536 + 16 * 4096, 0, d_tab, d_out, d_ticks, d_sink, sink_bytes, v.na, v.nb, v.nc, d_stop, v.nd, v.prio);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(probe, dim3(256), dim3(1024), 65536 + 16 * 4096, 0, d_tab, d_out, d_ticks, d_sink, sink_bytes, v.na, v.nb, v.nc, d_stop, v.nd, v.prio);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> tk(256 * 16);
        CK(hipMemcpy(tk.data(), d_ticks, tk.size() * 8, hipMemcpyDeviceToHost));
        double sum = 0; int n = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < v.na; ++w) { sum += (double)tk[b * 16 + w]; ++n; }
        const double cyc = sum / n / kSteps, ns = ms * 1e6 / kSteps;
        printf("%s %6.1f ns per step (launch %.0f us)   %6.1f s_memtime ticks per step   -> %.2f ticks per ns\n", v.name, ns, ms * 1e3, cyc, cyc / ns);
    }
    return 0;
}
