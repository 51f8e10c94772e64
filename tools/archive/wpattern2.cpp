// wpattern2.cpp — on-box probe (profiling only): does it matter whether records (4 B/row) and masked rows (2 B/row) go to two
// arrays or to one interleaved array?  65536 strings x 1024 rows, 1024 waves of 64 strings, 32-row steps like the storer.
//   mode 0: two arrays, string-major: rec[b][1024] u32 (128 B per step), msk[b][1024] u16 (128 B per two steps)
//   mode 1: one array, per string 64-row blocks of {256 B records, 128 B masked}: 384 contiguous bytes per two steps
//   mode 2: one array, per string {4 KiB records}{2 KiB masked}: same address set as mode 0 but one allocation / one stride
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE, int PADR = 0, int PADM = 0>
__global__ __launch_bounds__(256) void wk(char *A, char *Bm) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int s0 = lane >> 3, c = lane & 7;
    const uint4 v = make_uint4(lane, wave, 3, 4);
    for (int t = 0; t < 32; ++t) {
        for (int it = 0; it < 8; ++it) {  // records of this 32-row step: 8 strings x 128 B per instruction
            const size_t str = (size_t)wave * 64 + it * 8 + s0;
            char *p;
            if (MODE == 0) p = A + str * 4096 + t * 128 + c * 16;
            else if (MODE == 1) p = A + str * 6144 + (t >> 1) * 384 + (t & 1) * 128 + c * 16;
            else if (MODE == 2) p = A + str * 6144 + t * 128 + c * 16;
            else p = A + str * (size_t)(4096 + PADR) + t * 128 + c * 16;
            *(uint4 *)p = v;
        }
        if (t & 1) {
            for (int it = 0; it < 8; ++it) {  // masked rows of the 64-row block: 8 strings x 128 B per instruction
                const size_t str = (size_t)wave * 64 + it * 8 + s0;
                char *p;
                if (MODE == 0) p = Bm + str * 2048 + (t >> 1) * 128 + c * 16;
                else if (MODE == 1) p = A + str * 6144 + (t >> 1) * 384 + 256 + c * 16;
                else if (MODE == 2) p = A + str * 6144 + 4096 + (t >> 1) * 128 + c * 16;
                else p = Bm + str * (size_t)(2048 + PADM) + (t >> 1) * 128 + c * 16;
                *(uint4 *)p = v;
            }
        }
    }
}

template <int MODE, int PADR = 0, int PADM = 0> static void run(char *A, char *B) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((wk<MODE, PADR, PADM>), dim3(256), dim3(256), 0, 0, A, B);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 30; ++i) hipLaunchKernelGGL((wk<MODE, PADR, PADM>), dim3(256), dim3(256), 0, 0, A, B);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("mode %d pad %d/%d: %.1f us  %.2f TB/s\n", MODE, PADR, PADM, ms * 1e3 / 30, 65536.0 * 6144 / (ms / 30 * 1e-3) / 1e12);
}

int main() {
    char *A, *B; CK(hipMalloc(&A, (size_t)65536 * 8192)); CK(hipMalloc(&B, (size_t)65536 * 4096));
    for (int r = 0; r < 2; ++r) { run<0>(A, B); run<2>(A, B); run<3, 128, 128>(A, B); run<3, 128, 0>(A, B); run<3, 256, 128>(A, B); run<3, 512, 256>(A, B); run<3, 2048, 1024>(A, B); run<3, 1024, 0>(A, B); run<3, 0, 1024>(A, B); }
    return 0;
}
