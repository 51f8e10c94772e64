// hrx_host_walk.cpp — the native SMALL-BATCH host path of libhrx.so (SURVEY §8b "Who calls it": match_substrs hands
// over ONE string per call, src/lib.rs:316-318; a GPU launch for 1024 rows costs ~150 us of copies and synchronisation,
// the walk itself ~3 us on a host core).
//
// Same algorithm as the device lanes, not the oracle's: the walk goes through the dense fused table the library builds
// for the kernels (DefsSet::table_image, hrx_lane.h entry format), the reveal mask through the per-tile position
// bitvectors + carry-chain scans of hrx_lane.h (tile_masks, including the optimistic end-mask protocol and its fix-ups),
// 64 rows at a time, exactly what one GPU lane does for its string.  Nothing under oracle/ is linked, loaded or called.
//
//   walk            derive_states        src/lib.rs:804-823
//   tags / flags    derive_substr_ids, derive_is_start_end   src/lib.rs:825-888
//   rows, padding   match_substrs        src/lib.rs:339-348, 387-519
//   reveal mask     match_substrs        src/lib.rs:593-764
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

#include "hrx_defs.hpp"
#include "hrx_host_walk.hpp"
#include "hrx_lane.h"

namespace hrx {

uint64_t host_witness_one(const DefsSet &s, const uint8_t *chars, size_t n_raw, size_t M, uint32_t *records, uint16_t *masked) {
    const size_t D = s.defs.size();
    if (n_raw > M) return kStatusBadLength;
    const uint32_t n = (uint32_t)n_raw;
    const uint32_t *T = s.table_image.data();
    // per def: current fused entry (bits 10.. = absolute table row of the current state), first undefined transition
    std::vector<uint32_t> e(D), acc_state(D);
    std::vector<uint8_t> dead(D, 0);
    std::vector<uint32_t> err_pos(D, 0), err_state(D, 0), err_char(D, 0);
    for (size_t d = 0; d < D; ++d) {
        e[d] = s.consts[d].first_entry;              // states[d][0] = first_state_val: lib.rs:807
        acc_state[d] = s.consts[d].first_state;      // n == 0
    }
    MaskCarry mc = {0, 0, 0, 0};
    uint32_t sid_prev = 0, ov_row = 0xffffffffu;
    const uint32_t ntiles = (uint32_t)((M + 63) / 64);
    uint8_t sid_row[64];
    for (uint32_t t = 0; t < ntiles; ++t) {
        const uint32_t t0 = t * 64u;
        const uint32_t rows = (uint32_t)std::min<size_t>(64, M - t0);
        TileBits tb = {0, 0, 0};
        for (uint32_t p = 0; p < rows; ++p) {
            const uint32_t r = t0 + p;
            uint32_t sid = 0, stn = 0, enn = 0;
            for (size_t d = 0; d < D; ++d) {
                const DefConsts &c = s.consts[d];
                const uint32_t state = (e[d] >> kNextShift) - c.row_base;
                uint32_t tag = 0;
                if (r < n) {
                    const uint32_t ch = chars[r];
                    const uint32_t ne = T[(size_t)(e[d] >> kNextShift) * 256 + ch];     // delta(state, byte): lib.rs:810
                    if (ne >= c.dead_entry && !dead[d]) {                                 // lib.rs:817
                        dead[d] = 1;
                        err_pos[d] = r; err_state[d] = state; err_char[d] = ch;
                    }
                    tag = ne & kTagMask;
                    if (r + 1 >= M) tag &= ~kTagEnd;                                      // end_enable of row M-1 is never assigned: lib.rs:501
                    e[d] = ne;
                } else {
                    if (r == n) acc_state[d] = state;                                     // the state at row n: lib.rs:437-457
                    e[d] = c.dummy_entry;                                                 // rows > n: lib.rs:404-418
                }
                records[(size_t)r * D + d] = state | (tag << 16);
                sid += tag & 0xffu;
                stn += (tag >> 8) & 1u;
                enn += (tag >> 9) & 1u;
            }
            if (stn > 1) ov_row = std::min(ov_row, r);
            if (enn > 1) ov_row = std::min(ov_row, r + 1u);
            tb.st |= (uint64_t)(stn ? 1u : 0u) << p;
            tb.en1 |= (uint64_t)(enn ? 1u : 0u) << p;
            tb.ch |= (uint64_t)(sid != sid_prev ? 1u : 0u) << p;
            sid_prev = sid;
            sid_row[p] = (uint8_t)sid;
        }
        if (n == M && t + 1 == ntiles)   // n == M: row n does not exist, s[n] is the live state
            for (size_t d = 0; d < D; ++d) acc_state[d] = (e[d] >> kNextShift) - s.consts[d].row_base;
        // reveal masks: lib.rs:598-764
        const TileMasks tm = tile_masks<64>(tb, mc, t0, tile_is_exact(t0, n, (uint32_t)M), rows_below(t0, n));
        if (tm.fix)
            for (uint32_t r = tm.fix_start; r < t0; ++r) masked[r] = 0;
        for (uint32_t p = 0; p < rows; ++p) {
            const uint32_t r = t0 + p;
            masked[r] = ((tm.mask >> p) & 1u) ? (uint16_t)(chars[r] | (uint32_t)sid_row[p] << 8) : (uint16_t)0;   // lib.rs:752-761
        }
    }
    for (size_t d = 0; d < D; ++d)   // lowest def wins: the reference walks defs in order (lib.rs:806)
        if (dead[d]) return status_invalid((uint32_t)d, err_pos[d], err_state[d], err_char[d]);
    if (D > 1 && ov_row != 0xffffffffu) return status_overlap(ov_row);
    uint32_t accept = 0;
    for (size_t d = 0; d < D && d < 32; ++d) accept |= (acc_state[d] == s.consts[d].accepted_state ? 1u : 0u) << d;
    return status_ok(accept);
}

// The host threads of host_witness_batch: a process-wide pool of workers that sleep between jobs.  (Until round 6 every call started its threads anew — ~30 us apiece, one after the other:
// 254 threads for a 65536 x 1024 batch on a 256-core host cost as much as the walk itself, 8.9 ms per call where the pool takes it to what the host's memory gives.)  One job at a time
// (callers queue on job_mu: the shards of a multi-GPU call, clones on several threads); a job is a range of items handed out in order from an atomic counter, the calling thread works too.
// A forked child gets a pool of its own (get()).
class HostPool {
public:
    static HostPool &get() {
        // never destroyed: its workers sleep on it until the process ends.  A forked child (Python multiprocessing) inherits the object without its threads — and possibly with its
        // mutexes held: it gets a pool of its own and the inherited one is left alone (no destructor ever runs on thread handles that do not exist in this process).
        static std::mutex gm;
        static HostPool *p = nullptr;
        static pid_t owner = 0;
        std::lock_guard<std::mutex> g(gm);
        if (!p || owner != getpid()) { p = new HostPool(); owner = getpid(); }
        return *p;
    }
    // fn(lo, hi) over [0, n) in pieces of `grain`, on at most `threads` threads (the caller's included)
    void run(size_t n, size_t grain, size_t threads, const std::function<void(size_t, size_t)> &fn) {
        if (n == 0) return;
        if (grain == 0) grain = 1;
        const size_t pieces = (n + grain - 1) / grain;
        if (threads <= 1 || pieces <= 1) { fn(0, n); return; }
        std::lock_guard<std::mutex> job_lock(job_mu);
        const size_t helpers = ensure(std::min(threads - 1, pieces - 1));
        if (helpers == 0) { fn(0, n); return; }
        {
            std::lock_guard<std::mutex> lk(mu);
            job_fn = &fn; job_n = n; job_grain = grain; next.store(0); wanted = helpers; started = 0; finished = 0; ++generation;
        }
        cv_work.notify_all();
        work();
        std::unique_lock<std::mutex> lk(mu);
        wanted = started;                                   // (workers that have not picked the job up yet no longer may: nothing is left)
        cv_done.wait(lk, [&] { return finished == started; });
        job_fn = nullptr;
    }

private:
    std::mutex job_mu, mu;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> workers;
    const std::function<void(size_t, size_t)> *job_fn = nullptr;
    size_t job_n = 0, job_grain = 1, wanted = 0, started = 0, finished = 0;
    uint64_t generation = 0;
    std::atomic<size_t> next{0};

    void work() {
        for (;;) {
            const size_t lo = next.fetch_add(job_grain);
            if (lo >= job_n) return;
            (*job_fn)(lo, std::min(job_n, lo + job_grain));
        }
    }
    void loop() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_work.wait(lk, [&] { return generation != seen; });
            seen = generation;
            if (job_fn == nullptr || started >= wanted) continue;
            ++started;
            lk.unlock();
            work();
            lk.lock();
            if (++finished == started) cv_done.notify_one();
        }
    }
    // at least min(want, what the system gives) sleeping workers; returns how many there are (under job_mu)
    size_t ensure(size_t want) {
        while (workers.size() < want) {
            try {
                workers.emplace_back([this] { loop(); });
            } catch (const std::system_error &) {
                break;
            }
        }
        return std::min(workers.size(), want);
    }
};

void host_witness_batch(const DefsSet &s, const uint8_t *chars, size_t stride, const uint32_t *lens, size_t B, size_t M,
                        uint32_t *records, uint16_t *masked, uint64_t *status, int threads) {
    const size_t D = s.defs.size();
    auto run = [&](size_t lo, size_t hi) {
        for (size_t b = lo; b < hi; ++b)
            status[b] = host_witness_one(s, chars + b * stride, lens[b], M, records + b * M * D, masked + b * M);
    };
    if (threads <= 1 || B < 2) { run(0, B); return; }
    // pieces of ~32768 rows (~0.1-0.4 ms of walk): small enough to balance ragged strings over the threads, large enough that the counter is not contended
    const size_t grain = std::max<size_t>(1, 32768 / std::max<size_t>(1, M));
    HostPool::get().run(B, grain, (size_t)threads, run);
}

// derive_states (lib.rs:804-823) for one string: states[d * (n + 1) + i]; false + (state, char) of the reference's panic
bool host_derive_states(const DefsSet &s, const uint8_t *chars, size_t n, uint64_t *states, uint32_t &bad_state, uint32_t &bad_char) {
    const uint32_t *T = s.table_image.data();
    for (size_t d = 0; d < s.defs.size(); ++d) {
        const DefConsts &c = s.consts[d];
        uint32_t e = c.first_entry;
        states[d * (n + 1)] = c.first_state;
        for (size_t i = 0; i < n; ++i) {
            const uint32_t ne = T[(size_t)(e >> kNextShift) * 256 + chars[i]];
            if (ne >= c.dead_entry) {
                bad_state = (e >> kNextShift) - c.row_base;
                bad_char = chars[i];
                return false;
            }
            e = ne;
            states[d * (n + 1) + i + 1] = (e >> kNextShift) - c.row_base;
        }
    }
    return true;
}

// tags[d * n + i] = pair tag of (states[d][i], states[d][i+1]): substr id | is_start << 8 | is_end << 9 (lib.rs:825-888)
void host_pair_tags(const DefsSet &s, const uint64_t *states, size_t n, uint16_t *tags) {
    for (size_t d = 0; d < s.defs.size(); ++d) {
        const uint64_t ns = s.defs[d].allstr.largest_state_val + 1;
        for (size_t i = 0; i < n; ++i) {
            const uint64_t cur = states[d * (n + 1) + i], next = states[d * (n + 1) + i + 1];
            tags[d * n + i] = (cur < ns && next < ns) ? s.pair_tags[d][cur * ns + next] : (uint16_t)0;
        }
    }
}

// derive_is_start_end (lib.rs:847-888) for caller-supplied states AND substr ids: flags[d * n + i] bit0 is_start[d][i], bit1 is_end[d][i+1]
void host_endpoint_flags(const DefsSet &s, const uint64_t *states, const uint64_t *substr_ids, size_t n, uint8_t *flags) {
    for (size_t d = 0; d < s.defs.size(); ++d) {
        const uint64_t ns = s.defs[d].allstr.largest_state_val + 1, nsub = s.defs[d].substrs.size(), off = s.consts[d].substr_id_offset;
        for (size_t i = 0; i < n; ++i) {
            uint8_t f = 0;
            const uint64_t sid = substr_ids[d * n + i];
            if (sid != 0) {   // lib.rs:861-866, 874-879
                const uint64_t j = sid - off, cur = states[d * (n + 1) + i], next = states[d * (n + 1) + i + 1];
                if (j < nsub) {
                    if (cur < ns) f |= s.endpoint_member[d][j * ns + cur] & 1;
                    if (next < ns) f |= s.endpoint_member[d][j * ns + next] & 2;
                }
            }
            flags[d * n + i] = f;
        }
    }
}

}  // namespace hrx
