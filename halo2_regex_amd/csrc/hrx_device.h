// hrx_device.h — device-side helpers shared by the kernel translation units (hrx_kernel_sm.hip: string-major kernels,
// hrx_kernel_pm.hip: position-major kernel): LDS access by byte offset, the fused-table lookup, the per-lane walk state,
// the LDS ring hand-over between the waves of a pair, and the cached dynamic-LDS attribute.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>

#include "hrx_kernel.hpp"
#include "hrx_lane.h"

namespace hrx {

extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

// The kernels declare no static LDS, so the dynamic segment starts at LDS address 0 and a byte offset IS the
// LDS address: reads go through integer->address_space(3) casts so that no base add sits on the walk's
// dependent chain (witness_kernel traps if the assumption ever breaks).
typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
typedef uint32_t v2u32 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const v4u32 lds_cv4u32;
__device__ __forceinline__ uint32_t lds_u32(uint32_t off) { return *(lds_cu32 *)(uintptr_t)off; }
__device__ __forceinline__ uint4 lds_u128(uint32_t off) {
    const v4u32 v = *(lds_cv4u32 *)(uintptr_t)off;
    return make_uint4(v.x, v.y, v.z, v.w);
}

// one bit per non-zero byte, byte i -> bit i
__device__ __forceinline__ uint32_t nonzero_bytes4(uint32_t x) {
    const uint32_t nz = ((x | ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu)) >> 7) & 0x01010101u;
    return ((nz * 0x01020408u) >> 24) & 0xfu;
}

// delta lookup.  GTAB: the fused table did not fit the LDS budget and is read from global memory (it stays L2/MALL
// resident: every wave hammers the same few hundred KiB); same entry format, same byte offsets, ~10x the latency.
template <bool GTAB>
__device__ __forceinline__ uint32_t table_at(const WitnessArgs &a, uint32_t off) {
    if (GTAB) return a.table_image[off >> 2];
    return lds_u32(off);
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, s, 64));
    return v;
}

template <int D>
struct LaneRegs {
    uint32_t e[D];   // current fused entry of def d: bits 10.. = absolute table row of the CURRENT state
    uint32_t mx[D];  // running max of entries (reaching the dead row = an undefined transition)
    uint32_t sid_prev;
    uint32_t ov_row;  // D > 1: lowest row where two defs raise the same flag
};

constexpr uint32_t kNoFix = 0xffffffffu;

__device__ __forceinline__ uint32_t lds_vol_u32(uint32_t off) { return *(volatile lds_cu32 *)(uintptr_t)off; }
__device__ __forceinline__ void lds_store_u32(uint32_t off, uint32_t v) {
    *(volatile __attribute__((address_space(3))) uint32_t *)(uintptr_t)off = v;
}
// Workgroup -> slot in the launch's sequence of string groups.  Workgroups are dealt to the 8 XCDs round-robin, so with
// slot = workgroup index (the default) an XCD's walkers write the 4-KiB pieces of each position-major slab at byte offsets
// (4 KiB x XCD) mod 32 KiB.  In-kernel stamps (tools/state_probe4.py, profiles/r02_probes/xcd_*.txt) show the walkers of
// ODD workgroups finishing 5 us (fast per-process state) to 11 us (slow state) after those of even ones, launch after launch.
// kDbgXcdRemap deals each XCD a CONTIGUOUS eighth of the slots instead (its traffic = a contiguous eighth of every slab,
// spread evenly over all HBM stacks): the lag stays — it is not about which addresses an XCD writes — and the launch is
// 1.5 % slower (79.4 vs 78.1 us, six alternations over fresh processes), so it is off by default.  Needs gridDim.x % 8 == 0.
__device__ __forceinline__ uint32_t xcd_slot(const uint32_t wg, const uint32_t nwg, const bool remap) {
    return (remap && (nwg & 7u) == 0u) ? (wg & 7u) * (nwg >> 3) + (wg >> 3) : wg;
}

// 16 bytes per lane as a streaming (non-temporal) store: for output that is written once, in full lines, and never read
// by the kernel.  Inline asm on purpose (see hrx_walk_pm.h store16: LLVM merges an `if (nt)` diamond of a non-temporal and
// a plain store into one plain store); s_nop 1 = the two wait states a > 64-bit VMEM store needs on gfx940+ before a VALU
// may overwrite its data registers.
__device__ __forceinline__ void store16_nt(void *p, const uint4 &v) {
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(p), "v"(v4u32{v.x, v.y, v.z, v.w}) : "memory");
}

// wait until the tile counter at LDS offset `off` reaches `want`
__device__ __forceinline__ void ring_wait(uint32_t off, uint32_t want) {
    while ((int32_t)(lds_vol_u32(off) - want) < 0) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void ring_post(uint32_t off, uint32_t v) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // s_waitcnt lgkmcnt(0): the slot's LDS traffic is done
    lds_store_u32(off, v);
}

// ... for a counter that guards LDS data only: the LDS executes one wave's operations in the order they were issued, so the counter's store lands behind the wave's earlier
// LDS writes (and reads) without the wave waiting for them.
__device__ __forceinline__ void ring_post_lds(uint32_t off, uint32_t v) {
    asm volatile("" ::: "memory");
    lds_store_u32(off, v);
}
// ... when the counter was read ahead of time (`seen`, a volatile read some instructions back): no round trip if it had got there already
__device__ __forceinline__ void ring_wait_seen(uint32_t off, uint32_t want, uint32_t seen) {
    if ((int32_t)(seen - want) < 0) ring_wait(off, want);
    asm volatile("" ::: "memory");
}
// ... 16 bytes per lane at an SGPR base + a 32-bit lane offset
__device__ __forceinline__ void store16_nt_so(const void *sbase, uint32_t voff, const uint4 &v) {
    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" : : "v"(voff), "v"(v4u32{v.x, v.y, v.z, v.w}), "s"(sbase) : "memory");
}

// hipFuncSetAttribute costs several microseconds of host time: raise a kernel's dynamic-LDS limit only when a launch
// needs more than every earlier launch of that kernel did (the launch path is otherwise one hipLaunchKernelGGL).
// (Two shards of one device may race here — hrx_multi_* launches from one host thread per shard: the slow path is
// serialised and re-checks, so `granted` never runs ahead of the attribute actually set.)
template <class K>
static hipError_t ensure_lds(K k, std::atomic<size_t> &granted, size_t need) {
    if (need <= granted.load(std::memory_order_acquire)) return hipSuccess;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (need <= granted.load(std::memory_order_acquire)) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)need);
    if (e == hipSuccess) granted.store(need, std::memory_order_release);
    return e;
}

}  // namespace hrx
