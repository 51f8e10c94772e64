// csrc/hrx_arena_alloc.hpp: the offsets inside a placement arena.  A steady alloc / free churn (a prover's output buffers per batch) must be served from one arena
// for ever; ranges merge; nothing overlaps; a seeded random trace against a byte map.
#include <cstdio>
#include <cstdint>
#include <vector>

#include "../../halo2_regex_amd/csrc/hrx_arena_alloc.hpp"

static int failures = 0;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #c); ++failures; } } while (0)

int main() {
    using hrx::ArenaRanges;
    const size_t MiB = (size_t)1 << 20;
    {   // eight 256-MiB buffers fill a 2-GiB arena back to back, a ninth does not fit, the pieces come back merged
        ArenaRanges a(2048 * MiB);
        std::vector<size_t> offs;
        for (int i = 0; i < 8; ++i) { const size_t o = a.take(256 * MiB); CHECK(o == (size_t)i * 256 * MiB); offs.push_back(o); }
        CHECK(!a.fits(2 * MiB) && a.take(256 * MiB) == (size_t)-1 && a.live() == 8);
        CHECK(a.give(offs[3]) && a.give(offs[5]) && a.largest_free() == 256 * MiB);
        CHECK(a.give(offs[4]) && a.largest_free() == 768 * MiB);           // 3, 4, 5 merged
        CHECK(a.take(512 * MiB) == 3 * 256 * MiB && a.largest_free() == 256 * MiB);
        CHECK(!a.give(offs[4]) && !a.give(12345));                          // not the start of a live range (any more)
        for (size_t o : {offs[0], offs[1], offs[2], offs[6], offs[7], (size_t)(3 * 256 * MiB)}) CHECK(a.give(o));
        CHECK(a.live() == 0 && a.largest_free() == 2048 * MiB);
    }
    {   // the churn of tests/test_parity_gpu.py::test_one_context_per_thread_on_one_device, a million times: never exhausted
        ArenaRanges a(2048 * MiB);
        size_t live[4] = {a.take(256 * MiB), a.take(256 * MiB), a.take(256 * MiB), a.take(256 * MiB)};
        uint64_t x = 88172645463325252ull;
        for (int it = 0; it < 1000000; ++it) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            const int k = (int)(x & 3);
            CHECK(a.give(live[k]));
            live[k] = a.take(256 * MiB);
            if (live[k] == (size_t)-1) { CHECK(false); break; }
        }
        CHECK(a.live() == 4);
    }
    {   // seeded random trace of mixed sizes against a byte map (2-MiB units): no overlap, everything inside, full merge at the end
        const size_t U = 2 * MiB, N = 1024;
        ArenaRanges a(N * U);
        std::vector<int> owner(N, 0);
        struct Live { size_t off, len; int id; };
        std::vector<Live> live;
        uint64_t x = 0x9e3779b97f4a7c15ull;
        int next_id = 1;
        for (int it = 0; it < 200000; ++it) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            if ((x & 1) || live.empty()) {
                const size_t units = 1 + (size_t)((x >> 8) % 96);
                const size_t off = a.take(units * U);
                if (off == (size_t)-1) { CHECK(a.largest_free() < units * U); continue; }
                CHECK(off % U == 0 && off / U + units <= N);
                for (size_t u = off / U; u < off / U + units; ++u) { CHECK(owner[u] == 0); owner[u] = next_id; }
                live.push_back({off, units * U, next_id++});
            } else {
                const size_t k = (size_t)((x >> 8) % live.size());
                for (size_t u = live[k].off / U; u < (live[k].off + live[k].len) / U; ++u) { CHECK(owner[u] == live[k].id); owner[u] = 0; }
                CHECK(a.give(live[k].off));
                live[k] = live.back(); live.pop_back();
            }
            if (failures) break;
        }
        for (const Live &l : live) CHECK(a.give(l.off));
        CHECK(a.live() == 0 && a.largest_free() == N * U);
    }
    if (failures) return 1;
    std::puts("arena ranges: ok");
    return 0;
}
