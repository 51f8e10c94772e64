// hrx_place_api.cpp — the C ABI's placement-aware output allocation (hrx_alloc_output_pair, hrx_alloc_outputs_position_major, hrx_alloc_output_planes, hrx_device_free, hrx_probe_write_pair,
// hrx_ctx_set_placement, hrx_alloc_last_report) and the roofline diagnostics that share its measuring kernels (hrx_traffic_pass_device*: csrc/hrx_place.hip).  DESIGN.md §6.
#include "hrx_ctx.hpp"
#include "hrx_arena_alloc.hpp"
#include "hrx_place_rule.hpp"

#include <atomic>
#include <functional>

using namespace hrx;

extern "C" {

// Placement-aware allocation of the two output buffers (hrx_place.hip).
//
// Device memory is handed out top-down, so whatever a process allocates next lands right below what it allocated last — in
// the same class of the physical address space, where the launch's two write streams collide.  The search WALKS down the
// device memory instead and measures, with the two-stream probe, what lies there against where the records are:
//   * records >= kPlaceDirectFrom (1 GiB): masked-row candidates are allocated one after the other, each is measured against
//     the records buffer itself, and a rejected candidate stays allocated as the spacer that pushes the next one further
//     (round 2's scheme, now bounded by a budget and a relative acceptance rule);
//   * smaller records (the bench line's 256 MiB): buffers of that size live in the reach of the 256-MB Infinity Cache — a
//     probe over them measures the cache — and the driver's buddy allocator puts small blocks into whatever hole is highest,
//     not below the previous allocation.  They are therefore carved out of ARENAS: two 2-GiB blocks per context, one for
//     records and one for masked rows, the second found by walking 2-GiB blocks down the memory and measuring each, whole,
//     against the first (2 GiB per probe pass: the HBM regime).  Later requests are served from the same measured pair until
//     it is full; hrx_device_free returns a sub-buffer to its arena, and an arena is released when its last sub-buffer is
//     (and the context has moved on to another pair or is gone).
// Reference time: the same probe over two parts of ONE block (the records arena, or the records buffer): what two streams in
// one neighbourhood cost on this box.  A candidate is accepted when it is faster than that by kPlaceMargin — no absolute
// threshold (round 2's 6.9 TB/s did not hold on every box); failing that the fastest measured candidate is kept.  Spacers and
// rejected candidates are freed before the call returns.  The walk never takes more than kPlaceBudgetFrac of the free memory.
constexpr size_t kPlaceFromBytes = (size_t)128 << 20;    // below this the whole launch lives in the Infinity Cache: plain allocations
constexpr size_t kPlaceDirectFrom = (size_t)1 << 30;
constexpr size_t kPlaceArenaBytes = (size_t)2 << 30, kPlaceArenaAlign = (size_t)2 << 20;
constexpr double kPlaceBudgetFrac = 0.70;   // (the acceptance rule and its margins: hrx_place_rule.hpp)

struct hrx_place_arena {
    void *base = nullptr;
    int device = 0;
    hrx::ArenaRanges ranges;   // which offsets are handed out (first fit, freed ranges merge: hrx_arena_alloc.hpp) — an alloc / free churn is served from one pair for ever
    bool retired = false;      // no context serves requests from it any more: released with its last sub-buffer
    size_t pending_n = 0;      // of its live ranges, how many the caller has freed already (g_pending): not handed out again before the device has drained
};
static std::mutex g_arena_mu;
static std::map<uintptr_t, hrx_place_arena *> g_arena_of;   // sub-buffer -> arena (hrx_device_free has no context argument)
// DEFERRED REUSE.  hipFree waits for the device before the memory can be handed out again, and callers rely on that (a buffer may be freed while the launch that writes it is
// still in flight; the Python wrapper's finalizers do).  A freed arena range gets the same guarantee without a wait inside the free: it is parked here and becomes reusable
// only behind a device-wide wait that an ALLOCATION performs when its request does not fit otherwise (arena_reclaim_locked) — allocation is a synchronous call that is not legal
// inside a stream capture anyway, while a free may come from a finalizer thread at any time: a device-wide wait there stalled that thread for every stream of the device, and
// while another thread captured a graph in the global capture mode the runtime refused the wait (the free returned at once, the range was reusable with launches into it still
// in flight) or, in the relaxed mode, invalidated that capture (round 5's advisor; tests: test_freed_arena_range_is_not_reused_before_the_device_has_drained, test_arena_free_does_not_disturb_a_capture_in_another_thread).
static std::vector<std::pair<hrx_place_arena *, size_t>> g_pending;   // (arena, offset), under g_arena_mu

// under g_arena_mu: drops a's parked ranges (the arena itself goes: hipFree waits for the device)
static void arena_forget_pending(hrx_place_arena *a) {
    for (auto it = g_pending.begin(); it != g_pending.end();) it = it->first == a ? g_pending.erase(it) : it + 1;
    a->pending_n = 0;
}
// under g_arena_mu: one device-wide wait, then every parked range of that device is free again.  false: nothing was parked, or the wait was refused (a capture elsewhere)
static bool arena_reclaim_locked(int device) {
    bool any = false;
    for (const auto &e : g_pending) any = any || e.first->device == device;
    if (!any) return false;
    DeviceGuard guard;
    if (guard.set(device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return false; }
    for (auto it = g_pending.begin(); it != g_pending.end();) {
        if (it->first->device != device) { ++it; continue; }
        it->first->ranges.give(it->second);
        --it->first->pending_n;
        it = g_pending.erase(it);
    }
    return true;
}

// ranges / retired of an arena are only ever touched under g_arena_mu: hrx_device_free (any thread, no context argument — e.g. a finalizer
// while another thread allocates) releases sub-buffers concurrently with the owning context's takes.
static inline size_t arena_need(size_t bytes) { return (bytes + kPlaceArenaAlign - 1) / kPlaceArenaAlign * kPlaceArenaAlign; }
// a records and a masked-row sub-buffer out of the pair, or neither: the capacity check and both takes are ONE critical section
static bool arena_take_pair(hrx_place_arena *ra, size_t r_bytes, hrx_place_arena *ma, size_t m_bytes, void **r, void **m) {
    std::lock_guard<std::mutex> lk(g_arena_mu);
    auto fits = [&]() { return ra->ranges.fits(arena_need(r_bytes)) && ma->ranges.fits(arena_need(m_bytes)); };
    if (!fits() && !(arena_reclaim_locked(ra->device) && fits())) return false;
    const size_t ro = ra->ranges.take(arena_need(r_bytes)), mo = ma->ranges.take(arena_need(m_bytes));
    *r = (unsigned char *)ra->base + ro;
    *m = (unsigned char *)ma->base + mo;
    g_arena_of[(uintptr_t)*r] = ra;
    g_arena_of[(uintptr_t)*m] = ma;
    return true;
}
}  // extern "C"
void arena_retire(hrx_place_arena *a) {
    if (!a) return;
    std::lock_guard<std::mutex> lk(g_arena_mu);
    a->retired = true;
    if (a->ranges.live() == a->pending_n) { arena_forget_pending(a); (void)hipFree(a->base); delete a; }
}
extern "C" {
// true if ptr was a sub-buffer of an arena: its range is parked (DEFERRED REUSE above), nothing waits here
static bool arena_release(void *ptr) {
    std::lock_guard<std::mutex> lk(g_arena_mu);
    auto it = g_arena_of.find((uintptr_t)ptr);
    if (it == g_arena_of.end()) return false;
    hrx_place_arena *a = it->second;
    g_arena_of.erase(it);
    g_pending.emplace_back(a, (size_t)((unsigned char *)ptr - (unsigned char *)a->base));
    ++a->pending_n;
    if (a->retired && a->ranges.live() == a->pending_n) { arena_forget_pending(a); (void)hipFree(a->base); delete a; }     // (its last sub-buffer: hipFree waits for the device itself)
    return true;
}

// One measured arena pair per DEVICE and process, not per context: a prover that keeps one context per worker thread (a context serves one stream at a time)
// would otherwise walk once per context and hold 4 GiB of arenas in each.  mu serialises the walks and the replacement of a full pair; it is taken after the
// context's own mutex and before g_arena_mu (hrx_device_free takes only the latter).
struct hrx_place_pool {
    std::mutex mu;
    hrx_place_arena *rec = nullptr, *msk = nullptr;
    hrx_place_arena *stripe[4] = {nullptr, nullptr, nullptr, nullptr};   // the device's STRIPE ARENAS (hrx_alloc_output_planes of small buffers): stripe_n of them, mutually non-colliding
    size_t stripe_n = 0;
    hrx_place_report stripe_report{};
    hrx_place_report report{};     // of the walk that found the pair
    double seen_rate[HRX_MAX_DEFS + 1] = {};   // per number of defs D (the probe writes its two streams in the launch's ratio 4 D : 2, so rates of different D do not compare):
                                               // the fastest pairing any arena walk on this device has probed (bytes per microsecond)
    std::atomic<double> equal_level{0.0};      // choose_buffers' probe (two EQUAL streams): the mean of the faster level of pairings — across classes — in the last pool of this device that showed
                                               // two levels (bytes per microsecond; 0: none yet)
    int users = 0;                 // live contexts of the device (under g_arena_mu)
};
static std::map<int, hrx_place_pool *> g_pools;   // under g_arena_mu; entries are never removed (a few dozen bytes per device)
}  // extern "C"
hrx_place_pool *pool_acquire(int device) {
    std::lock_guard<std::mutex> lk(g_arena_mu);
    hrx_place_pool *&p = g_pools[device];
    if (!p) p = new hrx_place_pool();
    ++p->users;
    return p;
}
void pool_release(hrx_place_pool *p) {
    if (!p) return;
    hrx_place_arena *r = nullptr, *m = nullptr, *st[4] = {nullptr, nullptr, nullptr, nullptr};
    {
        std::lock_guard<std::mutex> pl(p->mu);
        std::lock_guard<std::mutex> lk(g_arena_mu);
        if (--p->users == 0) {
            r = p->rec; m = p->msk; p->rec = p->msk = nullptr; for (double &v : p->seen_rate) v = 0.0; p->report = hrx_place_report{};
            for (size_t i = 0; i < p->stripe_n; ++i) { st[i] = p->stripe[i]; p->stripe[i] = nullptr; }
            p->stripe_n = 0;
        }
    }
    arena_retire(r); arena_retire(m);
    for (hrx_place_arena *a : st) arena_retire(a);
}
extern "C" {

static void place_trace(const hrx_ctx *ctx, const char *fmt, ...) {
    if (!ctx->place_trace) return;
    va_list ap;
    va_start(ap, fmt);
    std::vfprintf(stderr, fmt, ap);
    va_end(ap);
}

// The walk.  A: the block everything is measured against (a_bytes), cand_bytes: the size of the blocks to walk with.  Returns the
// kept candidate (NULL: none could be allocated); everything else it allocated is freed.
static void *place_walk(hrx_ctx *ctx, void *A, size_t a_bytes, size_t cand_bytes, const bool arena_walk, const double seen_before, hrx_place_report &rep, double *best_rate_out) {
    const uint32_t D = (uint32_t)ctx->s.defs.size();
    unsigned long long *clk = (unsigned long long *)(ctx->d_group_counter + 4);   // 16 bytes of the context's 64-byte scratch word area
    size_t free_b = 0, total_b = 0;
    *best_rate_out = 0.0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    // never more than kPlaceBudgetFrac of what is free NOW.  The 2-GiB arena candidates of bench-sized outputs: 24 of them (48 GiB) as a rule — several contexts
    // or ranks on one device walk at the same time without pushing each other out of memory — and on only while NOTHING clearly above the same-block reference
    // has turned up (one lease of round 4: 24 candidates between 5.6 and 6.07 TB/s against a reference of 5.9, the bench line at 0.722 instead of 0.76; round 3's
    // unbounded walk had found a clear partner on every lease, up to ~100 candidates down), re-reading the free memory at every further step.
    // (arena_walk is the caller's statement, not inferred from the size: a direct walk whose masked buffer happens to measure 2 GiB — 1048576 x 1024 rows — keeps
    // the direct walk's caps.)  hrx_ctx_set_placement narrows the budget and adds a time cap; rep.capped says which bound ended the walk.
    size_t budget = (size_t)((double)free_b * kPlaceBudgetFrac);
    if (ctx->place_max_bytes) budget = std::min(budget, ctx->place_max_bytes);
    const int max_steps = arena_walk && !ctx->place_max_steps_set ? std::max(ctx->place_max_steps, hrx::kPlaceArenaHardSteps) : ctx->place_max_steps;
    rep.searched = 1;
    double ref_rate = 0.0;   // bytes per microsecond
    {   // the reference: both streams inside ONE block, in the launch's byte ratio (4 D : 2)
        const size_t a_rec = a_bytes / (4 * D + 2) * (4 * D) / 4096 * 4096;
        size_t wrote = 0;
        rep.ref_us = hrx::placement_probe_us(A, a_rec, (unsigned char *)A + a_rec, a_bytes - a_rec, D, ctx->stream, clk, &wrote);
        if (rep.ref_us > 0) ref_rate = (double)wrote / rep.ref_us;
        rep.ref_gbs = ref_rate * 1e-3;
    }
    std::vector<void *> spacers;       // rejected candidates: they are what pushes the next candidate further down
    void *best = nullptr;
    size_t spent = 0;
    double best_us = -1.0;
    hrx::PlaceWalk walk;               // the rates measured and when to stop: hrx_place_rule.hpp
    walk.ref_rate = ref_rate;
    walk.seen_before = seen_before;    // the fastest pairing earlier walks of the same kind measured (direct: this context's; arena: this device's)
    walk.arena = arena_walk;
    const auto t_walk = std::chrono::steady_clock::now();
    int i = 0;
    bool ended_by_rule = false;
    for (; i < max_steps; ++i) {
        if (spent + cand_bytes > budget) { rep.capped |= HRX_PLACE_CAPPED_BYTES; break; }
        if (!walk.may_take_another()) { ended_by_rule = true; break; }   // the arena soft cap: something clear of the reference is in hand
        if (arena_walk && i >= hrx::kPlaceArenaSoftSteps) {     // beyond it: leave other walkers / contexts of this device their share
            size_t f2 = 0, t2 = 0;
            if (hipMemGetInfo(&f2, &t2) != hipSuccess) { (void)hipGetLastError(); rep.capped |= HRX_PLACE_CAPPED_ALLOC; break; }
            if ((double)f2 < (1.0 - kPlaceBudgetFrac) * (double)t2) { rep.capped |= HRX_PLACE_CAPPED_BYTES; break; }
        }
        void *cand = nullptr;
        if (hipMalloc(&cand, cand_bytes) != hipSuccess) { (void)hipGetLastError(); rep.capped |= HRX_PLACE_CAPPED_ALLOC; break; }
        spent += cand_bytes;
        rep.peak_candidate_bytes = std::max(rep.peak_candidate_bytes, spent);
        const double us = hrx::placement_probe_us(A, a_bytes, cand, cand_bytes, D, ctx->stream, clk, &rep.probe_bytes);
        const double rate = us > 0 ? (double)rep.probe_bytes / us : 0.0;
        place_trace(ctx, "hrx placement: step %d candidate %p: %.1f us = %.2f TB/s, reference %.2f TB/s, %.1f ms into the walk\n", i, cand, us, rate * 1e-6, ref_rate * 1e-6,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_walk).count());
        ++rep.steps;
        if (i == 0) { rep.first_us = us; rep.first_gbs = rate * 1e-3; }
        const bool better = us >= 0 && (best_us < 0 || us < best_us);
        void *loser = better ? best : cand;
        if (better) { best = cand; best_us = us; rep.chosen_step = i; }
        if (loser) spacers.push_back(loser);
        walk.rates.push_back(rate);
        const double elapsed_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_walk).count();
        const hrx::PlaceVerdict v = walk.decide(elapsed_ms);
        if (v == hrx::PlaceVerdict::accept) { rep.accepted = 1; ended_by_rule = true; break; }
        if (v == hrx::PlaceVerdict::settle) {
            rep.accepted = walk.clear_of_reference() ? 1 : 0; ended_by_rule = true;
            if (elapsed_ms > (arena_walk ? hrx::kPlaceArenaHardMs : hrx::kPlaceHardMs)) rep.capped |= HRX_PLACE_CAPPED_TIME;   // the rule's own hard bound: whatever it holds
            break;
        }
        if (ctx->place_max_ms > 0 && elapsed_ms > ctx->place_max_ms) { rep.capped |= HRX_PLACE_CAPPED_TIME; break; }   // the caller's bound (hrx_ctx_set_placement)
    }
    if (i >= max_steps && !ended_by_rule) rep.capped |= HRX_PLACE_CAPPED_STEPS;
    const double best_rate = walk.best();
    *best_rate_out = best_rate;
    if (!rep.accepted && walk.clear_of_reference()) rep.accepted = 1;   // (a walk that ran into a cap with a pairing >= 10 % above the reference in hand)
    for (void *p : spacers) (void)hipFree(p);
    rep.best_us = best_us;
    rep.best_gbs = best_rate * 1e-3;
    place_trace(ctx, "hrx placement: kept step %d (%.1f us vs reference %.1f us, %s), %d steps\n", rep.chosen_step, best_us, rep.ref_us,
                rep.accepted ? "accepted" : "fastest measured", rep.steps);
    return best;
}

int hrx_alloc_output_pair(hrx_ctx *ctx, size_t records_bytes, size_t masked_bytes, void **records, void **masked) {
    if (!ctx || !records || !masked || records_bytes == 0 || masked_bytes == 0) return fail(HRX_ERR_ARG, "hrx_alloc_output_pair: bad argument");
    if (ctx->device == HRX_DEVICE_NONE) return fail(HRX_ERR_HIP, "host-only context (HRX_DEVICE_NONE): no device to allocate on");
    *records = nullptr; *masked = nullptr;
    std::lock_guard<std::mutex> lk(ctx->mu);   // the probe launches on the context's stream and uses its scratch
    DeviceGuard guard;
    HIP_TRY(guard.set(ctx->device));
    const auto t_begin = std::chrono::steady_clock::now();
    hrx_place_report rep{};
    auto done = [&](void *r, void *m) -> int {
        *records = r; *masked = m;
        rep.search_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        ctx->last_place = rep;
        return HRX_OK;
    };
    auto plain = [&]() -> int {
        void *r = nullptr, *m = nullptr;
        if (hipMalloc(&r, records_bytes) != hipSuccess) { (void)hipGetLastError(); return fail(HRX_ERR_HIP, "hrx_alloc_output_pair: out of device memory"); }
        if (hipMalloc(&m, masked_bytes) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(r); return fail(HRX_ERR_HIP, "hrx_alloc_output_pair: out of device memory"); }
        return done(r, m);
    };
    if (records_bytes < kPlaceFromBytes || !ctx->place_enabled) return plain();
    if (records_bytes >= kPlaceDirectFrom) {
        // ---- large outputs: candidates measured against the records buffer itself
        // (Round 5 also walked the RECORDS side — against one 12-GiB records buffer of cfg 4 all 48 masked-row candidates measure 6.4-6.8 TB/s, against the next one 5.8-6.3: where the records
        // lie sets the level — trying up to three records buffers and keeping the best pair: the allocation churn of 12-GiB spacers brought multi-second hipMalloc stalls (search_ms 3-4 s per
        // buffer set) and the launches did not follow the probe level closely enough to pay for it — cfg 4 at 0.655 with all three sets at 6.6-6.75.  Not kept.)
        void *rec = nullptr;
        if (hipMalloc(&rec, records_bytes) != hipSuccess) { (void)hipGetLastError(); return fail(HRX_ERR_HIP, "hrx_alloc_output_pair: out of device memory"); }
        double walked_best = 0.0;
        void *best = place_walk(ctx, rec, records_bytes, masked_bytes, /*arena_walk=*/false, ctx->place_seen_rate, rep, &walked_best);
        ctx->place_seen_rate = std::max(ctx->place_seen_rate, walked_best);
        if (!best && hipMalloc(&best, masked_bytes) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(rec); return fail(HRX_ERR_HIP, "hrx_alloc_output_pair: out of device memory"); }
        return done(rec, best);
    }
    // ---- bench-sized outputs: sub-buffers of the device's measured arena pair
    if (records_bytes > kPlaceArenaBytes || masked_bytes > kPlaceArenaBytes) return plain();
    hrx_place_pool *pool = ctx->pool;
    std::lock_guard<std::mutex> pl(pool->mu);
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (pool->rec && pool->msk) {
            void *r = nullptr, *m = nullptr;
            if (arena_take_pair(pool->rec, records_bytes, pool->msk, masked_bytes, &r, &m)) {
                if (attempt == 0) { rep = pool->report; rep.searched = 2; }   // served from the pair an earlier call (of any context of the device) measured
                return done(r, m);
            }
            arena_retire(pool->rec); arena_retire(pool->msk);     // full: a new pair
            pool->rec = pool->msk = nullptr;
        }
        void *A = nullptr;
        if (hipMalloc(&A, kPlaceArenaBytes) != hipSuccess) { (void)hipGetLastError(); return plain(); }
        // (the arena walks' own best rate per device and D: a direct walk's rate over other buffer sizes is not comparable — copied in here it could keep the accept
        // rule `b >= 0.96 seen()` from ever firing and every replacement pair walking to its caps with the pool mutex held)
        double &pool_seen = pool->seen_rate[std::min<size_t>(ctx->s.defs.size(), HRX_MAX_DEFS)];
        double walked_best = 0.0;
        void *X = place_walk(ctx, A, kPlaceArenaBytes, kPlaceArenaBytes, /*arena_walk=*/true, pool_seen, rep, &walked_best);
        pool_seen = std::max(pool_seen, walked_best);
        if (!X) { (void)hipFree(A); rep = hrx_place_report{}; return plain(); }
        pool->rec = new hrx_place_arena(); pool->rec->base = A; pool->rec->device = ctx->device; pool->rec->ranges.reset(kPlaceArenaBytes);
        pool->msk = new hrx_place_arena(); pool->msk->base = X; pool->msk->device = ctx->device; pool->msk->ranges.reset(kPlaceArenaBytes);
        pool->report = rep;
    }
    return plain();
}

int hrx_ctx_set_placement(hrx_ctx *ctx, int mode, size_t max_bytes, double max_ms) {
    if (!ctx) return fail(HRX_ERR_ARG, "NULL ctx");
    if (mode != HRX_PLACE_OFF && mode != HRX_PLACE_WALK) return fail(HRX_ERR_ARG, "hrx_ctx_set_placement: mode must be HRX_PLACE_OFF or HRX_PLACE_WALK");
    if (max_ms < 0) return fail(HRX_ERR_ARG, "hrx_ctx_set_placement: max_ms < 0");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->place_enabled = mode == HRX_PLACE_WALK;
    ctx->place_max_bytes = max_bytes;
    ctx->place_max_ms = max_ms;
    return HRX_OK;
}

int hrx_alloc_last_report(const hrx_ctx *ctx, hrx_place_report *out) {
    if (!ctx || !out) return fail(HRX_ERR_ARG, "NULL argument");
    *out = ctx->last_place;
    return HRX_OK;
}

int hrx_alloc_last_report_sized(const hrx_ctx *ctx, void *out, size_t out_bytes) {
    if (!ctx || !out) return fail(HRX_ERR_ARG, "NULL argument");
    std::memcpy(out, &ctx->last_place, std::min(out_bytes, sizeof ctx->last_place));      // a consumer built against an earlier header gets the fields it knows (the struct only ever grows at its end)
    return HRX_OK;
}

int hrx_traffic_pass_device(hrx_ctx *ctx, const uint8_t *chars, size_t stride, size_t B, size_t M, uint32_t *records, uint16_t *masked, void *stream) {
    if (!ctx) return fail(HRX_ERR_ARG, "NULL ctx");
    if (ctx->device == HRX_DEVICE_NONE) return fail(HRX_ERR_HIP, "host-only context (HRX_DEVICE_NONE): no device to launch on");
    if (B == 0) return HRX_OK;
    if (!chars || !records || !masked) return fail(HRX_ERR_ARG, "NULL buffer");
    if (M == 0 || M > (1u << 24) || B > 0xffffffffull - 64) return fail(HRX_ERR_ARG, "shape out of range");
    if ((stride & 15) || stride < 16 || ((uintptr_t)chars & 15) || ((uintptr_t)records & 15) || ((uintptr_t)masked & 15))
        return fail(HRX_ERR_ARG, "buffers must be 16-byte aligned with stride % 16 == 0 and stride >= 16");
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard guard;
    HIP_TRY(guard.set(ctx->device));
    // the store policy the planner gives the real launch of this shape
    WitnessArgs a{};
    a.layout = HRX_LAYOUT_POSITION_MAJOR | HRX_LAYOUT_INPUT_POSITION_MAJOR; a.B = (uint32_t)B; a.M = (uint32_t)M; a.D = (uint32_t)ctx->s.defs.size();
    LaunchInfo li{};
    li.split = 2;
    const uint32_t nt_mix = plan_nt_mix(a, li);
    HIP_TRY(launch_traffic_pass(chars, stride, B, M, a.D, records, masked, nt_mix, ctx->d_group_counter + 8, ctx->num_cus, (hipStream_t)stream));
    return HRX_OK;
}

int hrx_traffic_pass_device_layout(hrx_ctx *ctx, int layout, const uint8_t *chars, size_t stride, size_t B, size_t M, uint32_t *records, size_t rec_pitch,
                                   uint16_t *masked, size_t msk_pitch, void *stream) {
    if (layout == (HRX_LAYOUT_POSITION_MAJOR | HRX_LAYOUT_INPUT_POSITION_MAJOR)) return hrx_traffic_pass_device(ctx, chars, stride, B, M, records, masked, stream);
    if (layout != HRX_LAYOUT_STRING_MAJOR) return fail(HRX_ERR_ARG, "hrx_traffic_pass_device_layout: HRX_LAYOUT_STRING_MAJOR or HRX_LAYOUT_POSITION_MAJOR | HRX_LAYOUT_INPUT_POSITION_MAJOR");
    if (!ctx) return fail(HRX_ERR_ARG, "NULL ctx");
    if (ctx->device == HRX_DEVICE_NONE) return fail(HRX_ERR_HIP, "host-only context (HRX_DEVICE_NONE): no device to launch on");
    if (B == 0) return HRX_OK;
    if (!chars || !records || !masked) return fail(HRX_ERR_ARG, "NULL buffer");
    if (M == 0 || M > (1u << 24) || B > 0xffffffffull - 64) return fail(HRX_ERR_ARG, "shape out of range");
    if (rec_pitch == 0) rec_pitch = M;
    if (msk_pitch == 0) msk_pitch = M;
    const size_t D = ctx->s.defs.size();
    if (rec_pitch < M || msk_pitch < M || (M % 8) || (rec_pitch % 4) || (msk_pitch % 8) || rec_pitch > 0xffffffffull || msk_pitch > 0xffffffffull)
        return fail(HRX_ERR_ARG, "string-major traffic pass: M % 8 == 0, pitches >= M in multiples of 4 / 8 rows");
    if ((stride & 15) || stride < 16 || ((uintptr_t)chars & 15) || ((uintptr_t)records & 15) || ((uintptr_t)masked & 15))
        return fail(HRX_ERR_ARG, "buffers must be 16-byte aligned with stride % 16 == 0 and stride >= 16");
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard guard;
    HIP_TRY(guard.set(ctx->device));
    WitnessArgs a{};
    a.layout = HRX_LAYOUT_STRING_MAJOR; a.B = (uint32_t)B; a.M = (uint32_t)M; a.D = (uint32_t)D;
    a.rec_pitch = (uint32_t)rec_pitch; a.msk_pitch = (uint32_t)msk_pitch;      // (plan_nt_mix sizes the records by the pitch: without it the pass streamed every store where the launch writes some back — ADVICE r5)
    LaunchInfo li{};
    li.split = 1;
    const uint32_t nt_mix = plan_nt_mix(a, li);
    HIP_TRY(launch_traffic_pass_sm(chars, stride, B, M, (uint32_t)D, records, rec_pitch, masked, msk_pitch, nt_mix, ctx->d_group_counter + 8, ctx->num_cus, (hipStream_t)stream));
    return HRX_OK;
}

int hrx_traffic_pass_device_planes(hrx_ctx *ctx, const uint8_t *chars, size_t stride, size_t B, size_t M, uint32_t *const *record_planes, size_t n_planes,
                                   uint16_t *masked, void *stream) {
    if (!ctx) return fail(HRX_ERR_ARG, "NULL ctx");
    if (ctx->device == HRX_DEVICE_NONE) return fail(HRX_ERR_HIP, "host-only context (HRX_DEVICE_NONE): no device to launch on");
    if (B == 0) return HRX_OK;
    const size_t Dn = ctx->s.defs.size();
    if (!chars || !record_planes || !masked || !(n_planes == Dn || (Dn == 1 && n_planes == 2)) || n_planes > kMaxDefsPerLaunch) return fail(HRX_ERR_ARG, "NULL buffer, or not one plane per def (at most eight; one def: one buffer or two row stripes)");
    if (M == 0 || M > (1u << 24) || B > 0xffffffffull - 64) return fail(HRX_ERR_ARG, "shape out of range");
    if ((stride & 15) || stride < 16 || ((uintptr_t)chars & 15) || ((uintptr_t)masked & 15)) return fail(HRX_ERR_ARG, "buffers must be 16-byte aligned with stride % 16 == 0 and stride >= 16");
    for (size_t d = 0; d < n_planes; ++d)
        if (!record_planes[d] || ((uintptr_t)record_planes[d] & 15)) return fail(HRX_ERR_ARG, "record planes must be 16-byte aligned device buffers");
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard guard;
    HIP_TRY(guard.set(ctx->device));
    WitnessArgs a{};
    a.layout = HRX_LAYOUT_POSITION_MAJOR | HRX_LAYOUT_INPUT_POSITION_MAJOR; a.B = (uint32_t)B; a.M = (uint32_t)M; a.D = (uint32_t)Dn;
    LaunchInfo li{};
    li.split = 2;
    const uint32_t nt_mix = plan_nt_mix(a, li);
    HIP_TRY(launch_traffic_pass(chars, stride, B, M, a.D, record_planes[0], masked, nt_mix, ctx->d_group_counter + 8, ctx->num_cus, (hipStream_t)stream, record_planes, (uint32_t)(n_planes / Dn)));
    return HRX_OK;
}

int hrx_probe_write_pair(hrx_ctx *ctx, void *a, void *b, size_t bytes, double *gbs) {
    if (!ctx || !a || !b || !gbs || bytes < ((size_t)32 << 20)) return fail(HRX_ERR_ARG, "hrx_probe_write_pair: two device buffers of at least 32 MiB each");
    if (ctx->device == HRX_DEVICE_NONE) return fail(HRX_ERR_HIP, "host-only context (HRX_DEVICE_NONE): no device to measure on");
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard guard;
    HIP_TRY(guard.set(ctx->device));
    size_t wrote = 0;
    const double us = hrx::placement_probe_us(a, bytes, b, bytes, 0u, ctx->stream, (unsigned long long *)(ctx->d_group_counter + 4), &wrote);
    if (us <= 0) return fail(HRX_ERR_HIP, "hrx_probe_write_pair: the probe launch failed");
    *gbs = (double)wrote / us * 1e-3;
    return HRX_OK;
}

// Record planes + masked rows, each in a neighbourhood of its own: the launch's write streams spread over the classes of the physical address space instead of 4 D of
// its 4 D + 2 bytes per row going into one allocation.  A POOL of candidates — nrec + kPlanesSpare record-sized buffers, kPlanesMasked masked-row-sized ones, allocated one after the other (they
// walk down the device memory) — is measured pair by pair with the two-equal-streams probe (~1 ms per pair on the device clock), and the nrec record buffers + masked buffer whose BUSIEST CLASS
// takes the smallest share of the launch's output bytes (then: the fewest colliding pairings, the largest sum of pairings) are kept; the rest is freed before the call returns.  No absolute
// threshold: pairings in one class measure 5.2-6.2 TB/s, across classes 6.5-7.3 (profiles/r06_probes/plane_probe.txt, plane_select_cfg4.txt), and both levels move with the box: "colliding" =
// below the cut between the two levels of THIS pool.  Random draws of three 4-GiB planes + masked rows already run cfg 4's no-compute pass at 0.86 of peak in 54 of 60
// cases, against 0.65 for three planes of one class and 0.74-0.77 for the interleaved buffer: the selection only has to avoid the draws that collide.
constexpr size_t kPlanesSpare = 4, kPlanesMasked = 5, kPlanesGrow = 3, kPlanesMaxSets = 4096, kPlanesDry = 6;

}  // extern "C"

// nrec buffers of rec_bytes (each carrying w_rec of the launch's output bytes per row, in any unit) + one of msk_bytes (w_msk).  On success rec_out / msk_out own the kept buffers
// (hipMalloc'ed; everything else has been freed) and rep says what was measured; false: out of device memory (nothing is held).
// dry (optional): the REAL launch over a candidate set, milliseconds (< 0: could not run) — the best sets by the pairings' score are timed and the fastest is kept: the pairwise probe does
// not see everything (cfg 4: sets with the same pairings differ by 7 %, plane_select_cfg4.txt; a lease whose every set scored alike ran at 0.756 where others reach 0.80).
static bool choose_buffers(hrx_ctx *ctx, const size_t rec_bytes, const size_t nrec, const size_t msk_bytes, const size_t w_rec, const size_t w_msk, const bool walk,
                           std::vector<void *> &rec_out, void *&msk_out, hrx_place_report &rep,
                           const std::function<double(const std::vector<void *> &, void *)> &dry = nullptr) {
    const auto t_begin = std::chrono::steady_clock::now();
    std::vector<void *> pc, mc, spacers;     // record and masked-row candidates; blocks that only push the next candidates further down the memory
    auto free_all = [&]() { for (void *p : pc) if (p) (void)hipFree(p); for (void *p : mc) if (p) (void)hipFree(p); for (void *p : spacers) (void)hipFree(p); pc.clear(); mc.clear(); spacers.clear(); };
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
    size_t budget = (size_t)((double)free_b * kPlaceBudgetFrac), spent = 0;
    if (ctx->place_max_bytes) budget = std::min(budget, ctx->place_max_bytes);
    auto elapsed_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    auto take = [&](std::vector<void *> &v, size_t bytes, size_t must, size_t want) -> bool {      // `must` buffers or failure; up to `want` while memory and time allow
        while (v.size() < want) {
            const bool extra = v.size() >= must;
            if (extra && (spent + bytes > budget)) { rep.capped |= HRX_PLACE_CAPPED_BYTES; break; }
            if (extra && ctx->place_max_ms > 0 && elapsed_ms() > ctx->place_max_ms) { rep.capped |= HRX_PLACE_CAPPED_TIME; break; }
            void *p = nullptr;
            if (hipMalloc(&p, bytes) != hipSuccess) {
                (void)hipGetLastError();
                if (extra) { rep.capped |= HRX_PLACE_CAPPED_ALLOC; break; }
                return false;
            }
            v.push_back(p);
            spent += bytes;
        }
        rep.peak_candidate_bytes = std::max(rep.peak_candidate_bytes, spent);
        return true;
    };
    if (!take(pc, rec_bytes, nrec, walk ? nrec + kPlanesSpare : nrec) || !take(mc, msk_bytes, 1, walk ? kPlanesMasked : 1)) {
        free_all();
        return false;
    }
    std::vector<size_t> pick(nrec);
    for (size_t d = 0; d < nrec; ++d) pick[d] = d;
    size_t pick_m = 0;
    if (walk && (pc.size() > nrec || mc.size() > 1)) {
        rep.searched = 1;
        unsigned long long *clk = (unsigned long long *)(ctx->d_group_counter + 4);
        std::vector<std::vector<double>> pp, pm;       // bytes per microsecond of every pairing measured so far: record x record, masked x record
        auto probe = [&](void *x, void *y, size_t bytes) {
            size_t wrote = 0;
            const double us = hrx::placement_probe_us(x, bytes, y, bytes, 0u, ctx->stream, clk, &wrote);
            rep.probe_bytes = wrote;
            ++rep.steps;
            return us > 0 ? (double)wrote / us : 0.0;
        };
        double lo = 0, hi = 0, cut = 0;
        bool first = true, all_collide = false;
        int far_rounds = 0;
        struct Scored { size_t load, low; double sum, mn; std::vector<size_t> idx; size_t q; };
        std::vector<Scored> scored;
        size_t best_low = ~(size_t)0, best_load = ~(size_t)0;
        double best_sum = -1.0, best_min = 0.0;
        // what there is to find: with up to three buffers, a set in which nothing collides; with more, one whose busiest class takes a record buffer and the masked rows
        const size_t load_goal = nrec + 1 <= 3 ? std::max(w_rec, w_msk) : w_rec + w_msk;
        for (int round = 0;; ++round) {
            // ---- measure the pairings of the candidates that are new in this round
            const size_t P = pc.size(), Q = mc.size(), P0 = pp.size(), Q0 = pm.size();
            pp.resize(P);
            for (auto &r : pp) r.resize(P, 0.0);
            pm.resize(Q);
            for (auto &r : pm) r.resize(P, 0.0);
            for (size_t i = 0; i < P; ++i)
                for (size_t j = std::max(i + 1, P0); j < P; ++j) pp[i][j] = pp[j][i] = probe(pc[i], pc[j], rec_bytes);
            for (size_t q = 0; q < Q; ++q)
                for (size_t i = (q < Q0 ? P0 : 0); i < P; ++i) pm[q][i] = probe(mc[q], pc[i], std::min(msk_bytes, rec_bytes));      // (new columns of the old rows, whole new rows)
            // ---- two levels: pairings in one class of the address space and pairings across classes.  cut = halfway between the levels' means (two-means from the range's middle)
            std::vector<double> all;
            for (size_t i = 0; i < P; ++i) for (size_t j = i + 1; j < P; ++j) all.push_back(pp[i][j]);
            for (size_t q = 0; q < Q; ++q) for (size_t i = 0; i < P; ++i) all.push_back(pm[q][i]);
            lo = *std::min_element(all.begin(), all.end());
            hi = *std::max_element(all.begin(), all.end());
            cut = 0.5 * (lo + hi);
            for (int it = 0; it < 8; ++it) {
                double sl = 0, sh = 0; size_t nl = 0, nh = 0;
                for (double v : all) { if (v < cut) { sl += v; ++nl; } else { sh += v; ++nh; } }
                if (!nl || !nh) break;
                cut = 0.5 * (sl / nl + sh / nh);
            }
            double upper = 0.0;      // the mean of the faster level
            { double sh = 0; size_t nh = 0; for (double v : all) if (v >= cut) { sh += v; ++nh; } upper = nh ? sh / nh : hi; }
            cut = std::max(cut, 0.88 * hi);     // (one stalled probe — 4.3 TB/s among 5.7 .. 7.3 on one lease — must not drag the cut below the colliding level: pairings across
                                                //  classes lie within ~10 % of the fastest one)
            // Pairings that all lie at ONE level: nothing collides — or everything does (a pool of 15 GiB inside one class: classes are 8 to 64 GiB wide; seen on cfg 5's buffer sets on two
            // leases, every pairing 5.8-6.4 TB/s where the sets before it had 7.1-7.2 across classes, every launch over the pool 0.79-0.83 ms instead of 0.64-0.66).  Which of the two, only a
            // level measured elsewhere can say: the faster level of this device's last pool that showed both, failing that 6.9 TB/s (seen on every lease so far: 5.2-6.4 within a class,
            // 6.5-7.4 across).  A pool whose FASTER level is below 0.93 of that lies in one class: everything collides, and the next candidates come from behind a spacer.
            {
                const double known = ctx->pool ? ctx->pool->equal_level.load() : 0.0;
                all_collide = upper < 0.93 * (known > 0 ? known : 6.9e6);
                if (all_collide) cut = 2.0 * hi;
                else if (hi < 1.08 * lo) cut = 0.0;       // (nothing collides: only the sums rank)
                else if (ctx->pool) ctx->pool->equal_level.store(upper);
            }
            if (ctx->place_trace) {
                for (size_t i = P0; i < P; ++i) { std::string l; for (size_t j = 0; j < P; ++j) l += " " + std::to_string((int)(pp[i][j] * 1e-3)); place_trace(ctx, "hrx planes: record candidate %zu %p vs records (GB/s):%s\n", i, pc[i], l.c_str()); }
                for (size_t q = 0; q < Q; ++q) { std::string l; for (size_t j = 0; j < P; ++j) l += " " + std::to_string((int)(pm[q][j] * 1e-3)); place_trace(ctx, "hrx planes: masked candidate %zu %p vs records (GB/s):%s\n", q, mc[q], l.c_str()); }
            }
            // The score of a set: the largest share of the launch's output bytes that lands in one class — per buffer its own bytes per row (4 per plane, 2 for the masked rows)
            // plus those of every buffer it collides with, the maximum over the buffers — then the number of colliding pairings, then the sum of the pairings' rates.  cfg 3 over all 60
            // sets of 6 + 4 candidates (plane_select_cfg3.txt): nothing collides (largest share 4 of 10 bytes) 3.53 ms; the masked rows with a plane (6 of 10) 3.67-3.77; the two
            // planes with each other (8 of 10) 3.87-3.89; everything (10 of 10) 4.57-4.69 — the count of colliding pairings alone ranks the second and the third alike.  cfg 4 over all
            // 224 sets of 8 + 4 (plane_select_cfg4.txt): a plane and the masked rows in one class (6 of 14) 2.50-2.53 ms, two planes and the masked rows 2.64-2.88, everything 3.4-3.8.
            std::vector<size_t> idx(nrec);
            for (size_t d = 0; d < nrec; ++d) idx[d] = d;
            best_low = ~(size_t)0; best_sum = -1.0; best_load = ~(size_t)0;
            size_t sets = 0;
            scored.clear();
            for (;;) {
                for (size_t q = 0; q < Q; ++q) {
                    double mn = 1e30, sum = 0.0; size_t low = 0, load_m = w_msk, load_max = 0;
                    for (size_t x = 0; x < nrec; ++x) {
                        size_t load_x = w_rec;
                        for (size_t y = 0; y < nrec; ++y) {
                            if (y == x) continue;
                            const double v = pp[idx[x]][idx[y]];
                            if (v < cut) load_x += w_rec;
                            if (y > x) { mn = std::min(mn, v); sum += v; low += v < cut; }
                        }
                        const double v = pm[q][idx[x]]; mn = std::min(mn, v); sum += v; low += v < cut;
                        if (v < cut) { load_x += w_msk; load_m += w_rec; }
                        load_max = std::max(load_max, load_x);
                    }
                    load_max = std::max(load_max, load_m);
                    if (first) { first = false; rep.first_gbs = mn * 1e-3; }      // candidates 0 .. nrec - 1 + masked candidate 0: the plain-allocation draw
                    if (load_max < best_load || (load_max == best_load && (low < best_low || (low == best_low && sum > best_sum)))) {
                        best_load = load_max; best_low = low; best_sum = sum; best_min = mn; pick = idx; pick_m = q;
                    }
                    if (dry) scored.push_back(Scored{load_max, low, sum, mn, idx, q});
                }
                if (++sets >= kPlanesMaxSets) break;      // (more than three defs: the first sets in allocation order)
                size_t k = nrec;       // next combination
                while (k > 0 && idx[k - 1] == P - nrec + k - 1) --k;
                if (k == 0) break;
                ++idx[k - 1];
                for (size_t x = k; x < nrec; ++x) idx[x] = idx[x - 1] + 1;
            }
            place_trace(ctx, "hrx planes: round %d, %zu + %zu candidates: pairings %.2f .. %.2f TB/s, cut %.2f; best set: busiest class %zu of %zu output bytes per row, %zu colliding pairings, slowest %.2f TB/s\n",
                        round, P, Q, lo * 1e-6, hi * 1e-6, cut * 1e-6, best_load, nrec * w_rec + w_msk, best_low, best_min * 1e-6);
            // otherwise walk further down the memory, once or twice.  (Tried for the 2-GiB stripe arenas of one def's row stripes: four more rounds, each behind a 16-GiB spacer — 24 candidates
            // over ~120 GiB, 315 pairings, 3-4 s — still found no three mutually non-colliding arenas on a lease whose first 17 candidates showed two classes; not kept:
            // profiles/r06_probes/stripes_ab_cfg2.txt.)
            if (best_load <= load_goal || round >= (far_rounds ? 4 : 2)) break;
            if (all_collide) {
                // the whole pool in one class: the next candidates from further down the memory, behind a block that is allocated and never touched (at most a third of what the budget has left)
                size_t sp = std::min<size_t>((size_t)12 << 30, budget > spent ? (budget - spent) / 3 : 0) & ~(((size_t)1 << 30) - 1);
                void *p = nullptr;
                if (sp && hipMalloc(&p, sp) == hipSuccess) { spacers.push_back(p); spent += sp; ++far_rounds; place_trace(ctx, "hrx planes: every pairing collides: %zu GiB spacer before the next candidates\n", sp >> 30); }
                else (void)hipGetLastError();
            }
            const size_t before = pc.size();
            if (!take(pc, rec_bytes, before, before + kPlanesGrow) || pc.size() == before) break;
            (void)take(mc, msk_bytes, mc.size(), mc.size() + 1);      // (a masked-row candidate of the new neighbourhood as well)
        }
        // ---- the real launch over the best sets by score (at most kPlanesDry of them, no two sharing all their record buffers twice over): the fastest is kept
        if (dry && !scored.empty()) {
            std::sort(scored.begin(), scored.end(), [](const Scored &x, const Scored &y) { return x.load != y.load ? x.load < y.load : x.low != y.low ? x.low < y.low : x.sum > y.sum; });
            double best_ms = -1.0;
            size_t tried = 0;
            for (size_t k = 0; k < scored.size() && tried < kPlanesDry; ++k) {
                const Scored &c = scored[k];
                if (c.load > scored[0].load + std::max(w_rec, w_msk)) break;      // (only sets about as good as the best by score)
                std::vector<void *> rs;
                for (size_t d = 0; d < nrec; ++d) rs.push_back(pc[c.idx[d]]);
                const double ms = dry(rs, mc[c.q]);
                ++tried;
                if (ms <= 0) break;
                place_trace(ctx, "hrx planes: dry launch over set %zu (busiest class %zu, %zu colliding, sum %.1f TB/s): %.4f ms\n", k, c.load, c.low, c.sum * 1e-6, ms);
                if (tried == 1) rep.first_us = ms * 1e3;
                if (best_ms < 0 || ms < best_ms) { best_ms = ms; pick = c.idx; pick_m = c.q; best_load = c.load; best_low = c.low; best_min = c.mn; }
            }
            if (best_ms > 0) rep.best_us = best_ms * 1e3;
        }
        rep.ref_gbs = lo * 1e-3;           // the slowest pairing seen (the fastest is in the trace)
        rep.best_gbs = best_min * 1e-3;    // the kept set's slowest pairing
        rep.chosen_step = (int)best_load;  // ... and the output bytes per row that its busiest class takes (in units of the weights: planes 4 / masked rows 2; row stripes of one def 2 / 2)
        rep.accepted = (cut == 0.0 || best_load <= load_goal) ? 1 : 0;
    }
    rec_out.clear();
    for (size_t d = 0; d < nrec; ++d) { rec_out.push_back(pc[pick[d]]); pc[pick[d]] = nullptr; }
    msk_out = mc[pick_m];
    mc[pick_m] = nullptr;
    free_all();
    rep.search_ms = elapsed_ms();
    return true;
}

extern "C" {

void hrx_position_major_stripe_sizes(size_t B, size_t M, size_t n_stripes, size_t *stripe_u32, size_t *masked_u16) {
    if (n_stripes < 1) n_stripes = 1;
    if (stripe_u32) *stripe_u32 = ((M + 3) / 4 + n_stripes - 1) / n_stripes * B * 4;
    if (masked_u16) *masked_u16 = (M + 7) / 8 * B * 8;
}

// Small record buffers (below kPlaceDirectFrom each: the bench line's 128-MiB row stripes) are carved out of the device's STRIPE ARENAS — up to four 2-GiB blocks per device and process,
// mutually non-colliding as far as the box has them, found once with choose_buffers over 2-GiB candidates (a probe over buffers that fit the Infinity Cache would measure the cache); record
// buffer i comes out of arena i, the masked rows out of the last one; hrx_device_free returns a sub-buffer's range to its arena.
static bool stripe_arenas_take(hrx_ctx *ctx, const size_t nrec, const size_t rec_bytes, const size_t msk_bytes, std::vector<void *> &rec_out, void *&msk_out, hrx_place_report &rep) {
    hrx_place_pool *pool = ctx->pool;
    std::lock_guard<std::mutex> pl(pool->mu);
    const size_t need = nrec + 1;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (pool->stripe_n >= need) {
            std::lock_guard<std::mutex> lk(g_arena_mu);
            auto all_fit = [&]() {
                bool f = pool->stripe[need - 1]->ranges.fits(arena_need(msk_bytes));      // (the masked rows: the last arena of the set that was measured for `need` buffers)
                for (size_t i = 0; i < nrec && f; ++i) f = pool->stripe[i]->ranges.fits(arena_need(rec_bytes));
                return f;
            };
            const bool fits = all_fit() || (arena_reclaim_locked(ctx->device) && all_fit());      // (ranges the caller has freed come back behind a device-wide wait)
            if (fits) {
                rec_out.clear();
                for (size_t i = 0; i < nrec; ++i) {
                    void *p = (unsigned char *)pool->stripe[i]->base + pool->stripe[i]->ranges.take(arena_need(rec_bytes));
                    g_arena_of[(uintptr_t)p] = pool->stripe[i];
                    rec_out.push_back(p);
                }
                msk_out = (unsigned char *)pool->stripe[need - 1]->base + pool->stripe[need - 1]->ranges.take(arena_need(msk_bytes));
                g_arena_of[(uintptr_t)msk_out] = pool->stripe[need - 1];
                if (attempt == 0) { rep = pool->stripe_report; rep.searched = 2; }
                return true;
            }
        }
        // (full, or fewer arenas than this request needs: a new set)
        for (size_t i = 0; i < pool->stripe_n; ++i) arena_retire(pool->stripe[i]);
        pool->stripe_n = 0;
        std::vector<void *> ar;
        void *last = nullptr;
        hrx_place_report r2{};
        if (!choose_buffers(ctx, kPlaceArenaBytes, need - 1, kPlaceArenaBytes, 2, 2, true, ar, last, r2)) return false;
        ar.push_back(last);
        for (size_t i = 0; i < need; ++i) {
            hrx_place_arena *a = new hrx_place_arena();
            a->base = ar[i]; a->device = ctx->device; a->ranges.reset(kPlaceArenaBytes);
            pool->stripe[i] = a;
        }
        pool->stripe_n = need;
        pool->stripe_report = r2;
        rep = r2;
    }
    return false;
}

}  // extern "C"

// hrx_alloc_output_planes / hrx_alloc_output_planes_for_batch: `chars` != nullptr = the caller's batch (device pointers, `layout` as hrx_witness_batch_device_planes takes it) is what the dry
// launch runs
static int alloc_planes(hrx_ctx *ctx, size_t B, size_t M, size_t n_planes, uint32_t **record_planes, uint16_t **masked, int layout, const uint8_t *chars, size_t stride, const uint32_t *lens) {
    if (!ctx || !record_planes || !masked || B == 0 || M == 0) return fail(HRX_ERR_ARG, "hrx_alloc_output_planes: bad argument");
    if (ctx->device == HRX_DEVICE_NONE) return fail(HRX_ERR_HIP, "host-only context (HRX_DEVICE_NONE): no device to allocate on");
    const size_t D = ctx->s.defs.size();
    if (!(n_planes == D || (D == 1 && n_planes == 2))) return fail(HRX_ERR_ARG, "hrx_alloc_output_planes: one buffer per RegexDefs of the config (one def: one buffer, or two row stripes)");
    const size_t R = n_planes / D;
    size_t rec_u32 = 0, masked_u16 = 0;
    hrx_position_major_stripe_sizes(B, M, R, &rec_u32, &masked_u16);
    const size_t rec_bytes = rec_u32 * 4, masked_bytes = masked_u16 * 2;
    // one def, one buffer, no batch to measure with: the pair walk of hrx_alloc_output_pair.  (Tried: cfg 5's 2-GiB records + 1-GiB masked rows through the pool below — record AND masked-row
    // candidates, pairings, dry launch on a stand-in input: every kept set scored "nothing collides" and dry-launched at 0.61-0.63 ms, yet two of eight sets then ran the real batch at
    // 0.81-0.83 ms where the pair walk's sets run at 0.63-0.67 (profiles/r06_probes/cfg5_pool_dry.txt): with one def the INPUT is a seventh of the traffic, and where IT lies counts.)
    if (n_planes == 1 && !chars) return hrx_alloc_outputs_position_major(ctx, B, M, record_planes, masked);
    for (size_t d = 0; d < n_planes; ++d) record_planes[d] = nullptr;
    *masked = nullptr;
    std::lock_guard<std::mutex> lk(ctx->mu);   // the probe launches on the context's stream and uses its scratch
    DeviceGuard guard;
    HIP_TRY(guard.set(ctx->device));
    const auto t_begin = std::chrono::steady_clock::now();
    hrx_place_report rep{};
    std::vector<void *> rec;
    void *msk = nullptr;
    const size_t w_rec = 4 / R, w_msk = 2;
    const bool walk = ctx->place_enabled && rec_bytes * n_planes >= kPlaceFromBytes;
    bool ok = false;
    if (walk && rec_bytes < kPlaceDirectFrom && n_planes + 1 <= 4 && rec_bytes <= kPlaceArenaBytes / 2 && masked_bytes <= kPlaceArenaBytes / 2)
        ok = stripe_arenas_take(ctx, n_planes, rec_bytes, masked_bytes, rec, msk, rep);
    if (!ok) {
        rep = hrx_place_report{};
        // the dry launch: this context's own launch of B strings x M rows over a candidate set, on a constant input (every string M - 1 times the byte 'a': what the walk finds there does not
        // matter to where its bytes go), timed with events on the context's stream — one warm-up, the faster of two
        const bool direct_walk = walk && rec_bytes >= kPlaceDirectFrom / 4;
        size_t dstride = (M + 15) / 16 * 16;
        int dlayout = HRX_LAYOUT_POSITION_MAJOR | HRX_LAYOUT_INPUT_POSITION_MAJOR;
        DevBuf d_chars, d_lens, d_status;
        const uint8_t *dry_chars = chars;
        const uint32_t *dry_lens = lens;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        bool dry_ok = direct_walk && D <= kMaxDefsPerLaunch && ctx->place_dry;
        if (dry_ok && chars) {       // the caller's batch: what the buffers are for
            dstride = stride;
            dlayout = layout;
            dry_ok = d_status.reserve(B * 8) == hipSuccess;
        } else if (dry_ok) {
            size_t fb = 0, tb = 0;
            dry_ok = hipMemGetInfo(&fb, &tb) == hipSuccess && (double)(B * dstride) < 0.05 * (double)fb;      // (the input of the dry launch: at most a twentieth of what is free)
            dry_ok = dry_ok && d_chars.reserve(B * dstride) == hipSuccess && d_lens.reserve(B * 4) == hipSuccess && d_status.reserve(B * 8) == hipSuccess;
            dry_ok = dry_ok && hipMemsetAsync(d_chars.p, 'a', B * dstride, ctx->stream) == hipSuccess && hipMemsetD32Async((hipDeviceptr_t)d_lens.p, (int)(M - 1), B, ctx->stream) == hipSuccess;
            dry_chars = (const uint8_t *)d_chars.p;
            dry_lens = (const uint32_t *)d_lens.p;
        }
        if (dry_ok) dry_ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
        if (direct_walk && D <= kMaxDefsPerLaunch && ctx->place_dry && !dry_ok) (void)hipGetLastError();
        auto dry = [&](const std::vector<void *> &rs, void *mk) -> double {
            std::vector<uint32_t *> pl;
            for (void *p : rs) pl.push_back((uint32_t *)p);
            float best = -1.0f;
            for (int it = 0; it < 3; ++it) {
                if (hipEventRecord(e0, ctx->stream) != hipSuccess) return -1.0;
                if (launch_batch(ctx, dry_chars, dstride, dry_lens, B, M, pl[0], (uint16_t *)mk, (uint64_t *)d_status.p, ctx->stream, 0, 0, dlayout, pl.size() > 1 ? pl.data() : nullptr, pl.size()) != HRX_OK) return -1.0;
                if (hipEventRecord(e1, ctx->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) return -1.0;
                float t = 0.0f;
                if (hipEventElapsedTime(&t, e0, e1) != hipSuccess) return -1.0;
                if (it > 0 && (best < 0 || t < best)) best = t;
            }
            return (double)best;
        };
        ok = choose_buffers(ctx, rec_bytes, n_planes, masked_bytes, w_rec, w_msk, direct_walk, rec, msk, rep,
                            dry_ok ? std::function<double(const std::vector<void *> &, void *)>(dry) : std::function<double(const std::vector<void *> &, void *)>());
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        d_chars.release(); d_lens.release(); d_status.release();
    }
    if (!ok) return fail(HRX_ERR_HIP, "hrx_alloc_output_planes: out of device memory");
    for (size_t d = 0; d < n_planes; ++d) record_planes[d] = (uint32_t *)rec[d];
    *masked = (uint16_t *)msk;
    rep.search_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    ctx->last_place = rep;
    return HRX_OK;
}

extern "C" {

int hrx_alloc_output_planes(hrx_ctx *ctx, size_t B, size_t M, size_t n_planes, uint32_t **record_planes, uint16_t **masked) {
    return alloc_planes(ctx, B, M, n_planes, record_planes, masked, 0, nullptr, 0, nullptr);
}

int hrx_alloc_output_planes_for_batch(hrx_ctx *ctx, int layout, const uint8_t *chars, size_t stride, const uint32_t *lens, size_t B, size_t M, size_t n_planes, uint32_t **record_planes,
                                      uint16_t **masked) {
    if (!chars || !lens) return fail(HRX_ERR_ARG, "hrx_alloc_output_planes_for_batch: chars and lens are the batch the buffers are for (device pointers)");
    if (!(layout & HRX_LAYOUT_POSITION_MAJOR) || (layout & ~(HRX_LAYOUT_POSITION_MAJOR | HRX_LAYOUT_INPUT_POSITION_MAJOR)))
        return fail(HRX_ERR_ARG, "hrx_alloc_output_planes_for_batch: layout = HRX_LAYOUT_POSITION_MAJOR, optionally | HRX_LAYOUT_INPUT_POSITION_MAJOR");
    if ((stride & 15) || stride < 16 || ((uintptr_t)chars & 15) || ((uintptr_t)lens & 3)) return fail(HRX_ERR_ARG, "chars must be 16-byte aligned with stride % 16 == 0 and stride >= 16");
    return alloc_planes(ctx, B, M, n_planes, record_planes, masked, layout, chars, stride, lens);
}

int hrx_alloc_outputs_position_major(hrx_ctx *ctx, size_t B, size_t M, uint32_t **records, uint16_t **masked) {
    if (!ctx || !records || !masked || B == 0 || M == 0) return fail(HRX_ERR_ARG, "hrx_alloc_outputs_position_major: bad argument");
    size_t nr = 0, nm = 0;
    hrx_position_major_sizes(B, M, ctx->s.defs.size(), &nr, &nm);
    return hrx_alloc_output_pair(ctx, nr * 4, nm * 2, (void **)records, (void **)masked);
}

int hrx_device_free(void *ptr) {
    if (!ptr) return HRX_OK;
    if (arena_release(ptr)) return HRX_OK;   // a sub-buffer of a measured arena pair (hrx_alloc_output_pair)
    HIP_TRY(hipFree(ptr));
    return HRX_OK;
}


}  // extern "C"
