// hrx_host_api.cpp — host-buffer batches (hrx_witness_batch_host: what an unmodified caller of match_substrs' seam gets, src/lib.rs:311-318) and the multi-GPU driver
// (hrx_multi_*: shards by string index, no collective).  DESIGN.md §7, §9.
#include "hrx_ctx.hpp"
#include <sched.h>

#include <cmath>
#include <condition_variable>
#include <system_error>

#include "hrx_host_walk.hpp"

using namespace hrx;

int batch_host_locked(hrx_ctx *ctx, const uint8_t *chars, size_t stride, const uint32_t *lens, size_t B, size_t M,
                      uint32_t *records, uint16_t *masked, uint64_t *status, const bool one_stream) {
    if (B == 0) return HRX_OK;
    if (!chars || !lens || !records || !masked || !status) return fail(HRX_ERR_ARG, "NULL buffer");
    const size_t D = ctx->s.defs.size();
    const size_t dstride = stride ? (stride + 15) & ~(size_t)15 : 16;
    HIP_TRY(ctx->chars.reserve(dstride * B + 16));
    HIP_TRY(ctx->lens.reserve(4 * B));
    HIP_TRY(ctx->records.reserve(4 * B * M * D));
    HIP_TRY(ctx->masked.reserve(2 * B * M));
    HIP_TRY(ctx->status.reserve(8 * B));
    hipStream_t st = ctx->stream;
    // Large batches go CHUNK BY CHUNK, two host threads: a producer stages chunk c (host-to-device) and launches its walk on `stream`, the calling thread copies chunk
    // c - 1's finished rows out on `copy_stream` — the link is full duplex and a copy from or to pageable memory holds its host thread until it has landed, so one thread
    // cannot have both directions busy.  The rows leave 6 D' bytes per byte that comes in (448 MiB out, 64 MiB in at 65536 x 1024, D = 1): the copy out IS the call
    // (8.1 of round 4's 8.7 ms: ~55 GB/s, the link's rate in one direction); what the pipeline removes is the staging and the walk in front of it.
    const size_t out_per_string = M * (4 * D + 2);
    size_t cb = out_per_string ? (ctx->host_chunk_mib << 20) / out_per_string / 64 * 64 : B;   // ~48 MiB of rows per chunk (HRX_HOST_CHUNK_MIB when the context was created)
    if (cb < 1024) cb = 1024;
    const size_t nchunk = (B + cb - 1) / cb;
    // HRX_HOST_TRACE=1: one line per call on stderr — which way the call went, how long it took, per chunk when its input was on its way / its walk launched / its copy out began and ended
    const bool trace = ctx->host_trace;
    const auto t_call = std::chrono::steady_clock::now();
    auto ms_now = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count(); };
    // Pipelined or not.  On about every second box of this pool the pipelined call's device-to-host copies run at 25 GB/s instead of 55 for as long as the same call (or process) also
    // copies host-to-device on the other stream — 16.5 ms per 65536 x 1024 call instead of 8.0, while one stream doing in, walk, out takes 8.6 on every box and a plain copy of the same
    // bytes 7.2 (per-chunk traces, the variants tried: profiles/r05_probes/host_path_modes.txt).  So a context MEASURES: after its first big call (allocations, first touches) two calls go
    // pipelined and two on one stream, alternating; the faster way (the better of its two calls, per byte sent back; the pipeline unless the single stream is 10 % faster) takes the next
    // 62 calls, then the other way gets one call again; a pipelined call a quarter slower than the single stream's figure switches at once.  HRX_HOST_PIPELINE=1 / 0: always / never pipelined.
    // HRX_OPT_HOST_PIPELINE; the device part of a split call: one stream (8.6 ms per 65536 x 1024 on every box; the pipeline's 8.0 is 16.5 on every second one, and a measuring call
    // in the wrong mode made a split call look slower than the device alone: lease b of profiles/r06_leases/)
    const int force_pipe = one_stream ? 0 : ctx->host_pipeline == 1 ? 1 : ctx->host_pipeline == 2 ? 0 : -1;
    hrx_ctx::HostMode &hm = ctx->host_mode;
    const bool big = nchunk >= 3 && ctx->copy_stream != nullptr;
    bool sequential = !big, timed = false;
    if (big) {
        if (force_pipe >= 0) sequential = force_pipe == 0;
        else if (hm.calls == 0) sequential = false;                                      // not timed
        else if (hm.calls <= 4) { sequential = (hm.calls & 1u) == 0u; timed = true; }    // pipelined, one stream, pipelined, one stream
        else if (hm.until_probe == 0) { sequential = !hm.sequential; timed = true; }     // the other way's turn
        else { sequential = hm.sequential; timed = true; --hm.until_probe; }
    }
    auto account = [&](const bool was_sequential) {
        if (!big || force_pipe >= 0) return;
        const double ns_per_byte = ms_now() * 1e6 / (double)(B * out_per_string);
        const unsigned k = hm.calls++;
        if (!timed) return;
        double &fig = was_sequential ? hm.seq_ns_per_byte : hm.piped_ns_per_byte;
        fig = (k <= 4 && fig > 0.0) ? std::min(fig, ns_per_byte) : ns_per_byte;
        if (k < 4) return;
        if (k == 4 || was_sequential != hm.sequential) {       // a comparison is complete: decide
            hm.sequential = hm.seq_ns_per_byte < 0.9 * hm.piped_ns_per_byte;
            // the other way gets a call again after 62 — unless the comparison was lopsided (the slower way more than 1.5x: 16.5 against 8 ms where the pipeline's
            // copies run at half rate): then only after 1022, so that a long-lived prover does not pay a 2x call every 64th time
            const double a_ = hm.seq_ns_per_byte, b_ = hm.piped_ns_per_byte;
            hm.until_probe = (a_ > 0 && b_ > 0 && std::max(a_, b_) > 1.5 * std::min(a_, b_)) ? 1022 : 62;
        } else if (!was_sequential && ns_per_byte > 1.25 * hm.seq_ns_per_byte) {     // the box has changed its mind
            hm.sequential = true;
            hm.until_probe = 62;
        }
        if (trace) std::fprintf(stderr, "[hrx host] per byte sent back: pipelined %.4f ns, one stream %.4f ns -> %s\n", hm.piped_ns_per_byte, hm.seq_ns_per_byte, hm.sequential ? "one stream" : "pipelined");
    };
    if (sequential) {
        if (dstride != stride) HIP_TRY(hipMemsetAsync(ctx->chars.p, 0, dstride * B, st));
        if (stride) HIP_TRY(hipMemcpy2DAsync(ctx->chars.p, dstride, chars, stride, stride, B, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(ctx->lens.p, lens, 4 * B, hipMemcpyHostToDevice, st));
        if (int rc = launch_batch(ctx, (const uint8_t *)ctx->chars.p, dstride, (const uint32_t *)ctx->lens.p, B, M,
                                  (uint32_t *)ctx->records.p, (uint16_t *)ctx->masked.p, (uint64_t *)ctx->status.p, st))
            return rc;
        HIP_TRY(hipMemcpyAsync(records, ctx->records.p, 4 * B * M * D, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(masked, ctx->masked.p, 2 * B * M, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(status, ctx->status.p, 8 * B, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (trace) std::fprintf(stderr, "[hrx host] %zu strings on one stream: %.2f ms\n", B, ms_now());
        account(true);
        ctx->last_host.device_pipelined = 0;
        return HRX_OK;
    }
    std::vector<double> t_in(trace ? nchunk : 0), t_launch(trace ? nchunk : 0), t_out0(trace ? nchunk : 0), t_out1(trace ? nchunk : 0);
    std::vector<hipEvent_t> done(nchunk, nullptr);
    for (size_t c = 0; c < nchunk; ++c)
        if (hipEventCreateWithFlags(&done[c], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            for (hipEvent_t e : done) if (e) (void)hipEventDestroy(e);
            return fail(HRX_ERR_HIP, "hipEventCreate failed");
        }
    std::atomic<size_t> staged{0};
    std::atomic<int> prc{HRX_OK};
    std::string pmsg;
    const int device = ctx->device;
    std::mutex smu;                  // the consumer sleeps on `scv` until the chunk it waits for has been staged (it used to spin on `staged`: a core burnt per call)
    std::condition_variable scv;
    auto publish = [&](size_t v) { { std::lock_guard<std::mutex> g(smu); staged = v; } scv.notify_one(); };
    auto producer_body = [&] {
        DeviceGuard g2;
        if (g2.set(device) != hipSuccess) { pmsg = "hipSetDevice failed in the staging thread"; prc = HRX_ERR_HIP; publish(nchunk); return; }
        for (size_t c = 0; c < nchunk; ++c) {
            const size_t b0 = c * cb, n = std::min(cb, B - b0);
            unsigned char *dch = (unsigned char *)ctx->chars.p + b0 * dstride;
            hipError_t e = hipSuccess;
            if (dstride != stride) e = hipMemsetAsync(dch, 0, dstride * n, st);
            if (e == hipSuccess && stride) e = hipMemcpy2DAsync(dch, dstride, chars + b0 * stride, stride, stride, n, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipMemcpyAsync((uint32_t *)ctx->lens.p + b0, lens + b0, 4 * n, hipMemcpyHostToDevice, st);
            if (trace) t_in[c] = ms_now();
            int rc = HRX_OK;
            if (e == hipSuccess)
                rc = launch_batch(ctx, dch, dstride, (const uint32_t *)ctx->lens.p + b0, n, M, (uint32_t *)ctx->records.p + b0 * M * D,
                                  (uint16_t *)ctx->masked.p + b0 * M, (uint64_t *)ctx->status.p + b0, st);
            if (e == hipSuccess && rc == HRX_OK) e = hipEventRecord(done[c], st);
            if (trace) t_launch[c] = ms_now();
            if (e != hipSuccess || rc != HRX_OK) {
                pmsg = e != hipSuccess ? std::string("HIP error while staging a chunk: ") + hipGetErrorString(e) : std::string(hrx_last_error());
                (void)hipGetLastError();
                prc = e != hipSuccess ? HRX_ERR_HIP : rc;
                publish(nchunk);      // release the consumer
                return;
            }
            publish(c + 1);
        }
    };
    std::thread producer;
    try {
        producer = std::thread(producer_body);
    } catch (const std::system_error &) {      // no thread to be had: stage everything from this one, then copy out (nothing may unwind through the C ABI)
        producer_body();
    }
    int rc = HRX_OK;
    hipError_t ce = hipSuccess;
    for (size_t c = 0; c < nchunk && ce == hipSuccess; ++c) {
        {
            std::unique_lock<std::mutex> g(smu);
            scv.wait(g, [&] { return staged.load(std::memory_order_acquire) > c; });
        }
        if (prc.load() != HRX_OK) break;
        const size_t b0 = c * cb, n = std::min(cb, B - b0);
        if (trace) t_out0[c] = ms_now();
        ce = hipStreamWaitEvent(ctx->copy_stream, done[c], 0);
        if (ce == hipSuccess) ce = hipMemcpyAsync(records + b0 * M * D, (uint32_t *)ctx->records.p + b0 * M * D, 4 * n * M * D, hipMemcpyDeviceToHost, ctx->copy_stream);
        if (ce == hipSuccess) ce = hipMemcpyAsync(masked + b0 * M, (uint16_t *)ctx->masked.p + b0 * M, 2 * n * M, hipMemcpyDeviceToHost, ctx->copy_stream);
        if (ce == hipSuccess) ce = hipMemcpyAsync(status + b0, (uint64_t *)ctx->status.p + b0, 8 * n, hipMemcpyDeviceToHost, ctx->copy_stream);
        if (trace) t_out1[c] = ms_now();
    }
    if (producer.joinable()) producer.join();
    if (ce == hipSuccess) ce = hipStreamSynchronize(ctx->copy_stream);
    (void)hipStreamSynchronize(st);
    for (hipEvent_t e : done) (void)hipEventDestroy(e);
    if (trace) {
        std::string line = "[hrx host] " + std::to_string(nchunk) + " chunks of " + std::to_string(cb) + " strings, pipelined, ms: in/launched/out from-to";
        char buf[96];
        for (size_t c = 0; c < nchunk; ++c) { std::snprintf(buf, sizeof buf, " | %.2f/%.2f/%.2f-%.2f", t_in[c], t_launch[c], t_out0[c], t_out1[c]); line += buf; }
        std::snprintf(buf, sizeof buf, " | end %.2f\n", ms_now());
        line += buf;
        std::fputs(line.c_str(), stderr);
    }
    if (prc.load() == HRX_OK && ce == hipSuccess) account(false);
    ctx->last_host.device_pipelined = 1;
    if (prc.load() != HRX_OK) return fail(prc.load(), pmsg);
    if (ce != hipSuccess) { (void)hipGetLastError(); return fail(HRX_ERR_HIP, std::string("HIP error while copying a chunk out: ") + hipGetErrorString(ce)); }
    return rc;
}

extern "C" {

// How many host threads a walk may use: the cores of the calling thread's affinity mask (a rank pinned to its GPU's NUMA node walks with that node's cores, not with the machine's) — and
// no more than the CPU bandwidth the process's cgroup grants.  A container that sees 256 cores under a quota of 16 (cpu.max "1600000 100000": the GPU boxes of this pool) runs 256 threads in
// a burst and is then throttled for the rest of the period: the 65536 x 1024 walk took 3.7 ms on most calls and 83-92 ms on every third (profiles/r06_probes/host_routes_trace.txt).
static size_t cgroup_cpu_limit() {
    static const size_t limit = [] {
        auto read2 = [](const char *path, double &a, double &b) -> bool {
            FILE *f = std::fopen(path, "r");
            if (!f) return false;
            char x[64] = {0}, y[64] = {0};
            const int n = std::fscanf(f, "%63s %63s", x, y);
            std::fclose(f);
            if (n < 1 || std::strcmp(x, "max") == 0) return false;
            a = std::atof(x); b = n >= 2 ? std::atof(y) : 0.0;
            return true;
        };
        double quota = 0, period = 0;
        if (read2("/sys/fs/cgroup/cpu.max", quota, period) && quota > 0 && period > 0) return (size_t)std::max(1.0, std::ceil(quota / period));      // cgroup v2
        double q1 = 0, p1 = 0, unused = 0;
        if (read2("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", q1, unused) && q1 > 0 && read2("/sys/fs/cgroup/cpu/cpu.cfs_period_us", p1, unused) && p1 > 0)
            return (size_t)std::max(1.0, std::ceil(q1 / p1));                                                                                           // cgroup v1
        return (size_t)0;      // no limit
    }();
    return limit;
}
static size_t host_threads_available() {
    size_t n = std::max<size_t>(1, std::thread::hardware_concurrency());
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) n = (size_t)c; }
    const size_t lim = cgroup_cpu_limit();
    return lim ? std::min(n, lim) : n;
}

}  // extern "C"
int check_host_shape(size_t B, size_t M) {
    if (M == 0 || M > (1u << 24)) return fail(HRX_ERR_ARG, "max_chars_size must be in 1..2^24");
    if (B > 0xffffffffull - 64) return fail(HRX_ERR_ARG, "batch too large");
    return HRX_OK;
}
extern "C" {

// Batches of at least this many rows are SPLIT between the device and the host cores (HRX_HOST_ROUTE_AUTO): below it the launch + copies of a device part and the thread start-up of a host
// part are not small against either part's work
constexpr size_t kHostSplitFromRows = (size_t)1 << 22;

int hrx_witness_batch_host(hrx_ctx *ctx, const uint8_t *chars, size_t stride, const uint32_t *lens, size_t B, size_t M,
                           uint32_t *records, uint16_t *masked, uint64_t *status) {
    if (!ctx) return fail(HRX_ERR_ARG, "NULL ctx");
    if (B == 0) return HRX_OK;
    if (!chars || !lens || !records || !masked || !status) return fail(HRX_ERR_ARG, "NULL buffer");
    if (int rc = check_host_shape(B, M)) return rc;
    const size_t D = ctx->s.defs.size();
    const auto t_call = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    const size_t avail = ctx->host_threads > 0 ? (size_t)ctx->host_threads : host_threads_available();
    // the native walk over strings [b0, b0 + n): one host thread per ~8192 witness rows (~100 us of walk; a thread costs ~30 us to start), up to `threads`
    auto host_part = [&](size_t b0, size_t n, size_t threads) {
        const size_t want = std::max<size_t>(1, n * M / 8192);
        host_witness_batch(ctx->s, chars + b0 * stride, stride, lens + b0, n, M, records + b0 * M * D, masked + b0 * M, status + b0, (int)std::min(want, std::max<size_t>(1, threads)));
    };
    auto check_strides = [&]() -> int {
        for (size_t b = 0; b < B; ++b)
            if (lens[b] <= M && lens[b] > stride) return fail(HRX_ERR_ARG, "a string is longer than the stride");
        return HRX_OK;
    };
    // ---- which way.  A host-only context, a forced route, the debug flags of the tests (AUTO only), small batches: one way; otherwise both at once
    int route = ctx->host_route;
    if (ctx->device == HRX_DEVICE_NONE) route = HRX_HOST_ROUTE_HOST;
    else if (route == HRX_HOST_ROUTE_AUTO) {
        if (ctx->debug & kDbgNoHost) route = HRX_HOST_ROUTE_DEVICE;
        else if ((ctx->debug & kDbgForceHost) || B * M < ctx->host_threshold) route = HRX_HOST_ROUTE_HOST;
        else if (B * M < kHostSplitFromRows) route = HRX_HOST_ROUTE_DEVICE;
    }
    // ---- HRX_HOST_ROUTE_AUTO on a batch of at least kHostSplitFromRows rows: the fastest of THREE ways by this context's own measurements — everything through the device, everything on
    // the host cores, or both at once (strings [0, bs) through the device, [bs, B) on the host cores, bs from the rates of the two parts).  Which one wins is the host's: on a 256-core box
    // the host cores alone take 5.4 ms per 65536 x 1024 call, the device 8.0-8.7 (the copy back over the link), the split 6.9 (the parts slow each other down: the copies to and from
    // pageable memory and 254 walking threads share the host's memory); on a small host the device wins.  The context's calls 0 and 1 go through the device (0: allocations and first touches,
    // not recorded), 2 on the host cores, 3 and 4 split (half / half, then by the parts' rates); from then on the way with the smallest ns per row, whose figure every call refreshes; every
    // 32nd call re-measures one of the other two ways if its figure was within 1.5x of the best.
    enum { kDev = HRX_HOST_ROUTE_DEVICE, kHost = HRX_HOST_ROUTE_HOST, kSplit = 3 };
    int way = route == HRX_HOST_ROUTE_DEVICE ? kDev : route == HRX_HOST_ROUTE_HOST ? kHost : 0;
    const bool measured = way == 0;        // an AUTO call of a device context above the split threshold
    std::unique_lock<std::mutex> lk(ctx->mu, std::defer_lock);
    if (ctx->device != HRX_DEVICE_NONE) lk.lock();        // (the estimates, the report, the device part)
    hrx_ctx::HostRates &hr = ctx->host_rates;
    if (measured) {
        const unsigned k = hr.calls;
        if (k <= 1) way = kDev;
        else if (k == 2) way = kHost;
        else if (k <= 4) way = kSplit;
        else {
            const double d = hr.dev_alone, h = hr.host_alone, sp = hr.split_total;
            way = (d <= h && d <= sp) ? kDev : (h <= sp ? kHost : kSplit);
            if ((k % 32u) == 31u) {      // one of the other two gets a call again
                const double best = std::min(d, std::min(h, sp));
                const int other[2] = {way == kDev ? kHost : kDev, way == kSplit ? kHost : kSplit};
                const int cand = other[(k / 32u) & 1u];
                const double fig = cand == kDev ? d : cand == kHost ? h : sp;
                if (fig <= 1.5 * best) way = cand;
            }
        }
    }
    hrx_host_route_report &rep = ctx->last_host;
    auto finish = [&](int did, size_t bs, size_t hn, double dev_ms, double host_ms, size_t hthreads) {
        if (ctx->device == HRX_DEVICE_NONE) return;
        const double call_ms = ms_since(t_call), ns_row = call_ms * 1e6 / (double)(B * M);
        if (measured) {
            if (hr.calls > 0) {      // (call 0 pays for the allocations).  The mean of the old figure and the new one: one slow call does not dethrone a way
                auto upd = [&](double &fig) { fig = fig > 0 ? 0.5 * (fig + ns_row) : ns_row; };
                if (did == kDev) upd(hr.dev_alone);
                else if (did == kHost) upd(hr.host_alone);
                else {
                    if (hr.calls != 3u) upd(hr.split_total);      // (call 3 is the half / half split that only measures the two parts' rates: 11.9 ms where the split made from them takes 6.4)
                    // (mostly the new figure: a part's rate depends on its share, so the split has to follow quickly)
                    if (bs && dev_ms > 0) { const double m = dev_ms * 1e6 / (double)(bs * M); hr.dev_ns_per_row = hr.dev_ns_per_row > 0 ? 0.3 * hr.dev_ns_per_row + 0.7 * m : m; }
                    if (hn && host_ms > 0) { const double m = host_ms * 1e6 / (double)(hn * M); hr.host_ns_per_row = hr.host_ns_per_row > 0 ? 0.3 * hr.host_ns_per_row + 0.7 * m : m; }
                }
            }
            ++hr.calls;
        }
        rep.route = did == kSplit ? HRX_HOST_ROUTE_AUTO : did;
        rep.device_strings = bs; rep.host_strings = hn; rep.device_ms = dev_ms; rep.host_ms = host_ms; rep.call_ms = call_ms;
        rep.device_alone_ns_per_row = hr.dev_alone; rep.host_alone_ns_per_row = hr.host_alone; rep.split_ns_per_row = hr.split_total;
        rep.device_ns_per_row = hr.dev_ns_per_row; rep.host_ns_per_row = hr.host_ns_per_row; rep.host_threads = (int)hthreads;
        if (ctx->host_trace)
            std::fprintf(stderr, "[hrx host] %s: %zu strings through the device %.2f ms, %zu on %zu host threads %.2f ms, call %.2f ms; ns per row: device %.4f host %.4f split %.4f\n",
                         did == kDev ? "device" : did == kHost ? "host cores" : "split", bs, dev_ms, hn, hthreads, host_ms, call_ms, hr.dev_alone, hr.host_alone, hr.split_total);
    };
    if (way == kHost) {   // (the walk itself reads the context's tables only; a host-only context takes no lock and is re-entrant)
        if (int rc = check_strides()) return rc;
        const size_t nthreads = std::min(avail, std::max<size_t>(1, B * M / 8192));
        const auto t0 = std::chrono::steady_clock::now();
        if (lk.owns_lock() && !measured) lk.unlock();      // (a forced or small host call need not hold other callers of the context up)
        host_part(0, B, avail);
        const double host_ms = ms_since(t0);
        if (ctx->device != HRX_DEVICE_NONE && !lk.owns_lock()) lk.lock();
        finish(kHost, 0, B, 0.0, host_ms, nthreads);
        return HRX_OK;
    }
    DeviceGuard guard;
    HIP_TRY(guard.set(ctx->device));
    if (way == kDev) {
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = batch_host_locked(ctx, chars, stride, lens, B, M, records, masked, status);
        if (rc != HRX_OK) return rc;
        finish(kDev, B, 0, ms_since(t0), 0.0, 0);
        return HRX_OK;
    }
    // ---- both at once: all host cores but two (the staging thread and this one keep the device part fed)
    if (int rc = check_strides()) return rc;
    const double h = hr.host_ns_per_row, d = hr.dev_ns_per_row;
    double f_dev = (h > 0 && d > 0) ? h / (h + d) : 0.5;
    f_dev = std::min(15.0 / 16, std::max(1.0 / 16, f_dev));
    const size_t bs = std::max<size_t>(64, (size_t)((double)B * f_dev) / 64 * 64), hn = B - bs;
    const size_t hthreads = avail > 3 ? avail - 2 : 1;
    double host_ms = 0.0, dev_ms = 0.0;
    std::thread walker;
    bool walked = hn == 0;
    if (hn) {
        try {
            walker = std::thread([&] { const auto t0 = std::chrono::steady_clock::now(); host_part(bs, hn, hthreads); host_ms = ms_since(t0); });
        } catch (const std::system_error &) {      // (no thread to be had: the host part follows the device part on this thread)
        }
    }
    int rc = HRX_OK;
    {
        const auto t0 = std::chrono::steady_clock::now();
        rc = batch_host_locked(ctx, chars, stride, lens, bs, M, records, masked, status, /*one_stream=*/true);
        dev_ms = ms_since(t0);
    }
    if (walker.joinable()) { walker.join(); walked = true; }
    if (!walked) { const auto t0 = std::chrono::steady_clock::now(); host_part(bs, hn, avail); host_ms = ms_since(t0); }
    if (rc != HRX_OK) return rc;
    finish(kSplit, bs, hn, dev_ms, host_ms, hn ? hthreads : 0);
    return HRX_OK;
}

int hrx_ctx_host_route_report(const hrx_ctx *ctx, hrx_host_route_report *out, size_t out_bytes) {
    if (!ctx || !out) return fail(HRX_ERR_ARG, "NULL argument");
    std::memcpy(out, &ctx->last_host, std::min(out_bytes, sizeof ctx->last_host));      // (sized: the struct may grow at its end)
    return HRX_OK;
}

/* ------------------------------ multi-GPU driver ------------------------------ */

struct hrx_multi {
    std::vector<hrx_ctx *> ctxs;   // one per shard, in shard order
    size_t D = 0;
};

int hrx_multi_create(const hrx_defs *defs, const int *devices, int n_devices, hrx_multi **out) {
    if (!defs || !devices || !out || n_devices < 1) return fail(HRX_ERR_ARG, "NULL argument or no device");
    if (!defs->s.finalized) return fail(HRX_ERR_STATE, "call hrx_defs_finalize first");
    hrx_multi *m = new hrx_multi();
    m->D = defs->s.defs.size();
    for (int i = 0; i < n_devices; ++i) {
        hrx_ctx *c = nullptr;
        const int rc = hrx_ctx_create(defs, devices[i], &c);   // restores the caller's current device itself
        if (rc != HRX_OK) {
            hrx_multi_destroy(m);
            return rc;
        }
        m->ctxs.push_back(c);
    }
    // (the shards of one host-buffer call run at the same time: each walks with its share of the cores)
    for (hrx_ctx *c : m->ctxs) c->host_threads = (int)std::max<size_t>(1, host_threads_available() / (size_t)n_devices);
    *out = m;
    return HRX_OK;
}

void hrx_multi_destroy(hrx_multi *m) {
    if (!m) return;
    for (hrx_ctx *c : m->ctxs) hrx_ctx_destroy(c);
    delete m;
}

int hrx_multi_num_shards(const hrx_multi *m) { return m ? (int)m->ctxs.size() : 0; }
int hrx_multi_shard_device(const hrx_multi *m, int shard) { return m && shard >= 0 && shard < (int)m->ctxs.size() ? m->ctxs[(size_t)shard]->device : HRX_DEVICE_NONE; }
void *hrx_multi_shard_stream(const hrx_multi *m, int shard) { return m && shard >= 0 && shard < (int)m->ctxs.size() ? (void *)m->ctxs[(size_t)shard]->stream : nullptr; }

int hrx_multi_witness_batch_host(hrx_multi *m, const uint8_t *chars, size_t stride, const uint32_t *lens, size_t B, size_t M,
                                 uint32_t *records, uint16_t *masked, uint64_t *status) {
    if (!m) return fail(HRX_ERR_ARG, "NULL handle");
    if (B == 0) return HRX_OK;
    if (!chars || !lens || !records || !masked || !status) return fail(HRX_ERR_ARG, "NULL buffer");
    const int world = (int)m->ctxs.size();
    std::vector<int> rc((size_t)world, HRX_OK);
    std::vector<std::string> msg((size_t)world);
    std::vector<std::thread> th;
    for (int r = 0; r < world; ++r) {
        th.emplace_back([&, r] {
            size_t begin = 0, count = 0;
            hrx_shard_range(B, world, r, &begin, &count);
            if (count == 0) return;
            rc[(size_t)r] = hrx_witness_batch_host(m->ctxs[(size_t)r], chars + begin * stride, stride, lens + begin, count, M,
                                                   records + begin * M * m->D, masked + begin * M, status + begin);
            if (rc[(size_t)r] != HRX_OK) msg[(size_t)r] = hrx_last_error();   // thread-local: carry it to the caller's thread
        });
    }
    for (std::thread &t : th) t.join();
    for (int r = 0; r < world; ++r)
        if (rc[(size_t)r] != HRX_OK) return fail(rc[(size_t)r], "shard " + std::to_string(r) + ": " + msg[(size_t)r]);
    return HRX_OK;
}

int hrx_multi_witness_batch_device(hrx_multi *m, int layout, const uint8_t *const *chars, size_t stride, const uint32_t *const *lens,
                                   const size_t *counts, size_t M, uint32_t *const *records, uint16_t *const *masked,
                                   uint64_t *const *status) {
    if (!m) return fail(HRX_ERR_ARG, "NULL handle");
    if (!chars || !lens || !counts || !records || !masked || !status) return fail(HRX_ERR_ARG, "NULL argument");
    // one kernel per shard on the shard's own stream: the launches are asynchronous, so one host thread keeps all devices busy
    for (size_t r = 0; r < m->ctxs.size(); ++r) {
        if (counts[r] == 0) continue;
        hrx_ctx *c = m->ctxs[r];
        const int rc = hrx_witness_batch_device_layout(c, layout, chars[r], stride, lens[r], counts[r], M, records[r], masked[r], status[r],
                                                       (void *)c->stream);
        if (rc != HRX_OK) return fail(rc, "shard " + std::to_string(r) + ": " + std::string(hrx_last_error()));
    }
    return HRX_OK;
}

int hrx_multi_synchronize(hrx_multi *m) {
    if (!m) return fail(HRX_ERR_ARG, "NULL handle");
    for (hrx_ctx *c : m->ctxs) {
        if (c->device == HRX_DEVICE_NONE) continue;
        DeviceGuard guard;
        HIP_TRY(guard.set(c->device));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return HRX_OK;
}


}  // extern "C"
