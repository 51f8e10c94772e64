// hrx_ctx.hpp — what the translation units of the C ABI share (hrx_api.cpp: data model, contexts, the device entry points; hrx_describe_api.cpp, hrx_single_api.cpp: what their names say; hrx_place_api.cpp: placement-aware allocation and
// the roofline diagnostics; hrx_host_api.cpp: host-buffer batches and the multi-GPU driver; hrx_regex_api.cpp: definition generation): the handle structs, the error / device
// helpers and the few internal functions that cross the files.  Not installed, not part of include/hrx.h.
#pragma once
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hrx.h"
#include "hrx_defs.hpp"
#include "hrx_error.hpp"
#include "hrx_kernel.hpp"

#define HRX_INTERNAL __attribute__((visibility("hidden")))

static inline int fail(int code, const std::string &msg) { return hrx::set_last_error(code, msg); }

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) return fail(HRX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

// Entry points select the context's device for their HIP calls and restore the caller's current device on return
// (a torch process keeps its own notion of the current device).
struct DeviceGuard {
    int prev = -1;
    bool active = false;
    hipError_t set(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev == device) return hipSuccess;
        hipError_t e = hipSetDevice(device);
        active = (e == hipSuccess && prev >= 0);
        return e;
    }
    ~DeviceGuard() { if (active) (void)hipSetDevice(prev); }
};


struct hrx_defs {
    hrx::DefsSet s;
};

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = bytes + bytes / 4 + 4096;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};


// hrx_place_api.cpp
struct hrx_place_arena;
struct hrx_place_pool;
HRX_INTERNAL void arena_retire(hrx_place_arena *a);
HRX_INTERNAL hrx_place_pool *pool_acquire(int device);
HRX_INTERNAL void pool_release(hrx_place_pool *p);

struct hrx_ctx {
    hrx::DefsSet s;  // private copy: the ctx outlives / is independent of the hrx_defs it was made from
    int device = 0;          // HRX_DEVICE_NONE: no device, host walk only
    int num_cus = 0;
    uint32_t debug = 0;      // HRX_DEBUG_FLAGS, read once at creation (hrx_kernel.hpp)
    uint32_t tune = 0;       // hrx_ctx_set_option: kTune* bits (hrx_kernel.hpp)
    size_t host_threshold = HRX_DEFAULT_HOST_THRESHOLD;   // rows (B x M) below which host-buffer batches take the host walk
    hipStream_t stream = nullptr;
    // host-buffer batches of three chunks and more: pipelined (two streams) or one stream, whichever the last comparison on this box found faster (batch_host_locked)
    struct HostMode { unsigned calls = 0, until_probe = 0; bool sequential = false; double piped_ns_per_byte = 0.0, seq_ns_per_byte = 0.0; } host_mode;
    // hrx_witness_batch_host (hrx_host_api.cpp): the route (HRX_OPT_HOST_ROUTE), the host threads of the native walk (HRX_OPT_HOST_THREADS; 0: the cores the calling thread may run on), the
    // device part's transfer mode (HRX_OPT_HOST_PIPELINE: 0 measured, 1 pipelined, 2 one stream; HRX_HOST_PIPELINE in the environment of hrx_ctx_create sets the default), chunk size and trace
    int host_route = 0, host_threads = 0, host_pipeline = 0;
    size_t host_chunk_mib = 48;
    bool host_trace = false;
    // what the context's big AUTO calls measured, ns per row: each way alone, the split as a whole, and the split's two parts (both running at once: what the next split is made from)
    struct HostRates { double dev_alone = 0.0, host_alone = 0.0, split_total = 0.0, dev_ns_per_row = 0.0, host_ns_per_row = 0.0; unsigned calls = 0; } host_rates;
    hrx_host_route_report last_host{};
    hipStream_t copy_stream = nullptr;   // host-buffer batches: the device-to-host copies of finished chunks run here while the next chunks are staged and walked on `stream`
    uint32_t *d_table = nullptr;
    uint64_t *d_wide = nullptr;
    uint16_t *d_half = nullptr;
    uint8_t *d_pairtab = nullptr;
    uint8_t *d_bytetab = nullptr;
    std::vector<uint16_t *> d_pair;
    std::vector<uint8_t *> d_member;
    std::mutex mu;
    DevBuf chars, lens, records, masked, status, states, tags;
    // multi-pass configs (more than hrx::kMaxDefsPerPass defs, hrx_defs.hpp): per group the device images of its own DefsSet and its
    // private records / status buffers; one scratch array takes the passes' (meaningless) masked rows
    struct GroupDev {
        uint32_t *d_table = nullptr;
        uint64_t *d_wide = nullptr;
        uint16_t *d_half = nullptr;
        uint8_t *d_pairtab = nullptr;
        uint8_t *d_bytetab = nullptr;
        DevBuf records, status, summary;
    };
    std::vector<GroupDev> groups;
    struct CwGroupDev { DevBuf d_cw, status, summary; };
    std::vector<CwGroupDev> cw_groups;   // more than eight defs: the CW groups of DefsSet::cw_groups (position-major passes of up to eight defs each)
    DevBuf mp_masked;
    DevBuf mp_ov;          // multi-pass configs whose last pass merges the summaries: its cross-group overlap rows (WitnessArgs::merge_ov)
    bool mp_combine = false;   // HRX_MP_COMBINE=1: always the separate combine launch
    DevBuf tp_records, tp_masked;   // string-major callers served by the position-major path + transpose_pm_to_sm_kernel (hrx_kernel_tp.hip)
    DevBuf spec_cls, spec_ends, spec_fail, spec_init, spec_vstatus, spec_vinfo, spec_work;
    bool spec_qabs_ready = false;
    DevBuf spec_cimage;            // the scout's compact tables (class LUTs + class-indexed u16 tables), built with spec_qabs
    uint32_t spec_cimage_bytes = 0, spec_c_lut[hrx::kMaxDefsPerPass] = {}, spec_c_tab[hrx::kMaxDefsPerPass] = {}, spec_c_rowb[hrx::kMaxDefsPerPass] = {}, spec_c_inv[hrx::kMaxDefsPerPass] = {};
    uint32_t spec_qabs[hrx::kMaxDefsPerPass][8];   // chunked launches (hrx_kernel_spec.hip)
    // dynamic group assignment (hrx_kernel_pm.hip): a device counter, zeroed on the launch's stream right before the launch
    // (a memset node when the launches are captured into a HIP graph: replay-safe)
    uint32_t *d_group_counter = nullptr;
    // context-owned device scratch (the counter above, the group buffers of multi-pass configs) is shared by the launches of this
    // context: they must not overlap.  Launches on ONE stream are ordered anyway; a launch on another stream first waits (on the
    // host) for the stream that used the scratch last.  No events: these calls must stay legal inside a stream capture.
    hipStream_t scratch_stream = nullptr;
    bool scratch_used = false;
    // placement-aware output allocation (hrx_alloc_output_pair): tunables read once at creation, the last call's report
    bool place_enabled = true, place_trace = false, place_dry = true;     // (place_dry: HRX_OPT_PLACE_DRY_LAUNCH)
    int place_max_steps = 48;
    bool place_max_steps_set = false;   // HRX_PLACE_MAX_STEPS given: it bounds arena walks too (their own cap is kPlaceArenaHardSteps)
    double place_seen_rate = 0.0;     // bytes per microsecond of the best candidate any DIRECT walk (records >= 1 GiB) of this context has probed; arena walks keep theirs per device (hrx_place_pool)
    size_t place_max_bytes = 0;       // hrx_ctx_set_placement: the most device memory a walk may hold at once (0: 70 % of what is free)
    double place_max_ms = 0.0;        // ... and the wall-clock time a walk may take (0: the rule's own bounds, hrx_place_rule.hpp)
    hrx_place_report last_place{};
    struct hrx_place_pool *pool = nullptr;   // bench-sized outputs: the device's measured arena pair, shared by every context of that device in this process
    DevBuf d_cw;                    // CLASS-WIDE image of a config of 4 .. 7 defs (DefsSet::cw_image): the single-launch def-parallel path
#ifdef HRX_STAMPS
    DevBuf stamps;                  // tools-only build: 8 u64 per walker pair of the position-major kernel (hrx_kernel_pm.hip)
#endif
};

// device copies of one DefsSet's kernel-side images

// hrx_host_api.cpp
HRX_INTERNAL int check_host_shape(size_t B, size_t M);
// hrx_api.cpp: one batch on the context's device (device pointers; the caller holds ctx->mu and has selected the device)
HRX_INTERNAL int launch_batch(hrx_ctx *ctx, const uint8_t *chars, size_t stride, const uint32_t *lens, size_t B, size_t M, uint32_t *records, uint16_t *masked, uint64_t *status,
                              hipStream_t st, size_t rec_pitch = 0, size_t msk_pitch = 0, int layout = 0, uint32_t *const *planes = nullptr, size_t n_planes = 0);
// hrx_host_api.cpp: a host-buffer batch through the device (staged, walked, copied back); the caller holds ctx->mu and has selected the device
// (one_stream: in, walk, out on one stream whatever HRX_OPT_HOST_PIPELINE says and without taking part in the context's comparison of its two transfer modes: the device part of a split call)
HRX_INTERNAL int batch_host_locked(hrx_ctx *ctx, const uint8_t *chars, size_t stride, const uint32_t *lens, size_t B, size_t M, uint32_t *records, uint16_t *masked, uint64_t *status,
                                   bool one_stream = false);
