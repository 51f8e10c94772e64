// hostcopy_probe.cpp — on-box probe: how to move 384 MiB of results into a caller's PAGEABLE buffer fastest.
//   (a) hipMemcpy D2H straight into pageable memory   (b) hipHostRegister + hipMemcpy + hipHostUnregister
//   (c) D2H into pinned staging chunks + memcpy by N threads, pipelined
//   hipcc -O3 -std=c++17 tools/hostcopy_probe.cpp -o /tmp/hostcopy_probe -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t N = 384u << 20;
    char *d; CK(hipMalloc(&d, N)); CK(hipMemset(d, 1, N));
    char *h = (char *)aligned_alloc(4096, N); memset(h, 0, N);
    for (int rep = 0; rep < 2; ++rep) {
        double t = now(); CK(hipMemcpy(h, d, N, hipMemcpyDeviceToHost)); printf("(a) pageable hipMemcpy: %.1f ms\n", (now() - t) * 1e3);
    }
    for (int rep = 0; rep < 2; ++rep) {
        double t = now(); CK(hipHostRegister(h, N, hipHostRegisterDefault)); double t1 = now();
        CK(hipMemcpy(h, d, N, hipMemcpyDeviceToHost)); double t2 = now(); CK(hipHostUnregister(h)); double t3 = now();
        printf("(b) register %.1f + copy %.1f + unregister %.1f = %.1f ms\n", (t1 - t) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t) * 1e3);
    }
    const size_t CH = 16u << 20; const int NB = 4;
    char *p[NB]; for (int i = 0; i < NB; ++i) CK(hipHostMalloc((void **)&p[i], CH, hipHostMallocDefault));
    hipStream_t s; CK(hipStreamCreate(&s)); hipEvent_t ev[NB]; for (int i = 0; i < NB; ++i) CK(hipEventCreate(&ev[i]));
    for (int nth : {1, 4, 8}) {
        double t = now();
        const size_t nch = N / CH;
        for (size_t c = 0; c < nch + NB - 1; ++c) {
            if (c < nch) { CK(hipMemcpyAsync(p[c % NB], d + c * CH, CH, hipMemcpyDeviceToHost, s)); CK(hipEventRecord(ev[c % NB], s)); }
            if (c >= NB - 1) {
                const size_t k = c - (NB - 1);
                CK(hipEventSynchronize(ev[k % NB]));
                std::vector<std::thread> th;
                for (int i = 0; i < nth; ++i) th.emplace_back([&, i] { const size_t part = CH / nth; memcpy(h + k * CH + i * part, p[k % NB] + i * part, part); });
                for (auto &x : th) x.join();
            }
        }
        printf("(c) pinned staging, %d memcpy threads: %.1f ms\n", nth, (now() - t) * 1e3);
    }
    return 0;
}
