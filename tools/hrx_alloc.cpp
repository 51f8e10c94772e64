// hrx_alloc.cpp — TOOLS ONLY (linked into libhrx_ablation.so, not into libhrx.so; not declared in include/hrx.h):
// hrx_chunked_alloc / hrx_chunked_free, device buffers assembled from 2-MiB physical chunks, for the placement probes
// (tools/alloc_policy_probe.py, tools/set_probe3.py).
//
// Background.  On an MI355X the bandwidth of concurrent write streams depends on where the streams lie in the PHYSICAL
// address space: 6.3 TB/s for two streams inside one region, 7.4-7.5 TB/s across two (tools/halves_probe.cpp on one
// contiguous 128-GiB allocation).  A witness launch over several GB writes two such streams — records and masked rows —
// and with hipMalloc'ed buffers its time is an accident of placement: 0.97 / 1.02 / 1.07 / 1.13 / 1.18 ms for 262144 x 2048 B
// at D = 2, one level per position-major block whose records collide with its masked rows; the state belongs to the
// records allocation for as long as it lives, follows neither the virtual address nor offsets inside an allocation, and
// is gone when the masked-row stores are skipped (tools/set_probe4.py, tools/alloc_kind_probe.py, tools/va_align_probe.py,
// tools/state_ablation.py).  This allocator was the attempt to turn the accident into a policy: a virtual range (HIP
// virtual memory management) over 2-MiB physical allocations created one by one, in batches mapped round-robin, optionally
// with spacer allocations between the batches.  Outcome: 0.91-0.95 ms in every process of one box session (better than the
// best accident), 1.13 ms in the next session, and the Infinity-Cache-resident bench line loses 3-5 % — placement still
// decides, the allocator only changes which accident one gets.  So it stays a probe; the product takes caller-owned
// buffers as they come.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../include/hrx.h"

extern "C" int hrx_chunked_alloc(int device, size_t bytes, void **out);
extern "C" int hrx_chunked_free(void *ptr);

namespace {

struct Mapping {
    size_t bytes = 0;      // mapped size (a multiple of the chunk size); 0: a plain hipMalloc
    size_t chunk = 0;
    int device = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
};
std::mutex g_mu;
std::unordered_map<void *, Mapping> g_maps;

constexpr size_t kChunk = (size_t)2 << 20;
constexpr size_t kSmall = (size_t)64 << 20;      // below this an ordinary allocation: nothing to spread
constexpr size_t kPositions = 4;                 // batches the chunks are created in; consecutive chunks come from different batches

void release(void *p, Mapping &m, size_t mapped_chunks) {
    if (mapped_chunks) (void)hipMemUnmap(p, mapped_chunks * m.chunk);
    for (auto h : m.handles) (void)hipMemRelease(h);
    if (p) (void)hipMemAddressFree(p, m.bytes);
}

}  // namespace

extern "C" int hrx_chunked_alloc(int device, size_t bytes, void **out) {
    if (!out || bytes == 0) return HRX_ERR_ARG;
    *out = nullptr;
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return HRX_ERR_HIP; }
    size_t positions = kPositions, spacer = 0, chunk = kChunk;
    if (const char *e = getenv("HRX_ALLOC_POSITIONS")) positions = (size_t)atol(e);
    if (const char *e = getenv("HRX_ALLOC_SPACER_MIB")) spacer = (size_t)atol(e) << 20;
    if (const char *e = getenv("HRX_ALLOC_CHUNK_MIB")) chunk = (size_t)atol(e) << 20;
    Mapping m;
    m.device = device;
    void *p = nullptr;
    bool ok = false;
    if (bytes >= kSmall && positions >= 1) {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        const size_t n = (bytes + chunk - 1) / chunk;
        positions = std::min(positions, n);
        m.bytes = n * chunk;
        m.chunk = chunk;
        std::vector<void *> spacers;     // (probe builds only: allocations held between the batches)
        std::vector<std::vector<hipMemGenericAllocationHandle_t>> pool(positions);
        ok = hipMemAddressReserve(&p, m.bytes, chunk, nullptr, 0) == hipSuccess;
        if (!ok) p = nullptr;
        for (size_t k = 0; ok && k < positions; ++k) {
            const size_t want = n / positions + (k < n % positions ? 1 : 0);
            for (size_t c = 0; ok && c < want; ++c) {
                hipMemGenericAllocationHandle_t h;
                ok = hipMemCreate(&h, chunk, &prop, 0) == hipSuccess;
                if (ok) { pool[k].push_back(h); m.handles.push_back(h); }
            }
            if (ok && k + 1 < positions && spacer) {
                void *s = nullptr;
                if (hipMalloc(&s, spacer) == hipSuccess) spacers.push_back(s);   // (no spacer: the next batch simply lands next to this one)
                else (void)hipGetLastError();
            }
        }
        for (void *s : spacers) (void)hipFree(s);
        size_t mapped = 0;
        for (size_t c = 0; ok && c < n; ++c) {
            ok = hipMemMap((char *)p + c * chunk, chunk, 0, pool[c % positions][c / positions], 0) == hipSuccess;
            if (ok) mapped = c + 1;
        }
        if (ok) {
            hipMemAccessDesc d = {};
            d.location = prop.location;
            d.flags = hipMemAccessFlagsProtReadWrite;
            ok = hipMemSetAccess(p, m.bytes, &d, 1) == hipSuccess;
        }
        if (!ok) { (void)hipGetLastError(); release(p, m, mapped); p = nullptr; m = Mapping(); m.device = device; }
    }
    if (!ok) {   // small, or the virtual-memory path failed (no room, an old runtime): an ordinary allocation
        ok = hipMalloc(&p, bytes) == hipSuccess;
        if (!ok) (void)hipGetLastError();
    }
    (void)hipSetDevice(prev);
    if (!ok) return HRX_ERR_HIP;
    std::lock_guard<std::mutex> lk(g_mu);
    g_maps.emplace(p, std::move(m));
    *out = p;
    return HRX_OK;
}

extern "C" int hrx_chunked_free(void *ptr) {
    if (!ptr) return HRX_OK;
    Mapping m;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_maps.find(ptr);
        if (it == g_maps.end()) return HRX_ERR_ARG;
        m = std::move(it->second);
        g_maps.erase(it);
    }
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(m.device);
    (void)hipDeviceSynchronize();
    if (m.bytes == 0) (void)hipFree(ptr);
    else release(ptr, m, m.bytes / m.chunk);
    (void)hipSetDevice(prev);
    return HRX_OK;
}
