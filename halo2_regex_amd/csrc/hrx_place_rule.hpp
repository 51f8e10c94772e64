// hrx_place_rule.hpp — when the placement walk of hrx_alloc_output_pair stops (hrx_api.cpp place_walk).  Pure arithmetic on the probe rates
// measured so far, kept apart from the walk so that it can be replayed on a host without a device: tests/host_cpp/test_place_rule.cpp feeds it the candidate
// sequences the round's leases recorded (profiles/r04_probes/cfg5_batch_sweep.txt, profiles/r04_config_sweep_new_rule/).
#pragma once
#include <algorithm>
#include <vector>

namespace hrx {

constexpr double kPlaceMargin = 1.10;      // accepted: >= 10 % more bytes per microsecond than two streams inside one block (colliding pairs: +-4 %, clear ones: +20-25 %)
constexpr double kPlaceNearBest = 0.96;    // ... and within 4 % of the best pairing any walk of this context has measured
constexpr double kPlaceAsSeen = 0.97;      // ... at once if within 3 % of what an EARLIER walk of this context kept
constexpr double kPlaceWalkMs = 250.0;     // a walk past its eighth candidate that HAS something clear of the reference — and as good as what the context has seen — ends after a quarter of a second (arena candidates: a second)
constexpr double kPlaceHardMs = 2000.0, kPlaceArenaHardMs = 8000.0;   // ... and any walk after this, whatever it holds (one lease of round 4 took 66 ms per 2-GiB hipMalloc where the others take 2:
                                                                      // the one-second bound ended its arena walk at 17 colliding candidates and the bench line ran at 0.70 instead of 0.76)
constexpr int kPlaceMinCandidates = 4;     // the median of fewer says nothing: clear pairings are about one in eight
constexpr int kPlaceMinCandidatesFirst = 8;   // ... and the FIRST direct walk of a context (nothing seen before to hold a candidate against) looks at eight: against a same-block reference of the
                                              // hard kind (4.3-4.7 TB/s) a middle-kind candidate (5.9-6.5) clears reference and median by 10 % as well as a clear one (6.8-7.2) does — round 5, lease b:
                                              // cfg 3's first buffer set kept a 5.9 after four candidates, the second matched it, 0.697 where a process of its own on the same lease found 6.8 / 7.0 and ran at 0.736
constexpr int kPlaceArenaSoftSteps = 24, kPlaceArenaHardSteps = 96;   // 2-GiB arena candidates: 24 as a rule, on only while nothing clear of the reference is in hand

enum class PlaceVerdict { go_on, accept, settle };   // settle: stop with the fastest candidate measured (accepted only if it is clear of the reference)

struct PlaceWalk {
    double ref_rate = 0.0;        // the same-block reference (any unit, the same as the candidates')
    double seen_before = 0.0;     // the fastest pairing EARLIER walks of the context measured (0: none)
    bool arena = false;           // 2-GiB arena candidates (bench-sized outputs)
    std::vector<double> rates;    // of every candidate measured so far, in walk order (0 = a failed probe)

    double best() const { double b = 0.0; for (double r : rates) b = std::max(b, r); return b; }
    double worst() const { double w = 0.0; for (double r : rates) if (r > 0 && (w == 0.0 || r < w)) w = r; return w; }
    double seen() const { return std::max(seen_before, best()); }
    // the lower middle of the candidates measured (two clear ones among four must not hide each other)
    double median() const {
        std::vector<double> s;
        for (double r : rates) if (r > 0) s.push_back(r);
        if (s.empty()) return 0.0;
        std::sort(s.begin(), s.end());
        return s[(s.size() - 1) / 2];
    }
    bool clear_of_reference() const { return ref_rate > 0 && best() >= kPlaceMargin * ref_rate; }

    // before candidate number rates.size() is allocated: may the walk go past the arena soft cap?  (the caller also re-reads the free memory there)
    bool may_take_another() const { return !(arena && (int)rates.size() >= kPlaceArenaSoftSteps && clear_of_reference()); }

    // after a candidate has been measured (its rate is the last element of rates); elapsed_ms: since the walk began
    PlaceVerdict decide(const double elapsed_ms) const {
        const int i = (int)rates.size() - 1;
        const double b = best();
        if (i < 0 || b <= 0 || ref_rate <= 0) return PlaceVerdict::go_on;
        // Candidates come in kinds — pairings that collide (4.7 and 5.7-6.1 TB/s on the probe: two such kinds on some boxes) and clear ones (6.9-7.3), about one
        // in eight.  The walk ends when the fastest candidate so far is clearly (>= 10 %) above BOTH the same-block reference and the MEDIAN candidate seen — at
        // least four candidates, so that the median is a colliding one: round 3's rule (10 % above the SLOWEST of at least two) took a 6.1 for clear next to a 4.7
        // and cost cfg 5 at 393216 x 4096 a fifth of its rate — and within 4 % of the best pairing any walk of this context has measured (a later buffer set must
        // not settle for less than the first one found: cfg 5, 0.365 -> 0.417 ms per step on such a box).
        const int min_cand = (!arena && seen_before <= 0.0) ? kPlaceMinCandidatesFirst : kPlaceMinCandidates;
        if (i + 1 >= min_cand && b >= kPlaceMargin * std::max(ref_rate, median()) && b >= kPlaceNearBest * seen()) return PlaceVerdict::accept;
        // ... or as soon as it is as good as the pairing an EARLIER walk of this context kept (where most neighbours are clear the median rule never fires: 48 steps
        // and 1.6 s for one buffer set of cfg 5 seen)
        if (i >= 1 && seen_before > 0 && b >= kPlaceAsSeen * seen_before && clear_of_reference()) return PlaceVerdict::accept;
        // ... or when it has cost too much: allocating and freeing candidates of several GiB takes tens of milliseconds each — soon if something clear of the
        // reference is in hand, late if not (a one-time cost of seconds against ~8 % of every launch on those buffers)
        if (i >= 7 && elapsed_ms > (arena ? 4.0 * kPlaceWalkMs : kPlaceWalkMs) && clear_of_reference() && b >= kPlaceNearBest * seen()) return PlaceVerdict::settle;
        if (elapsed_ms > (arena ? kPlaceArenaHardMs : kPlaceHardMs)) return PlaceVerdict::settle;
        // no kinds on this box / for this pair of sizes: ten candidates within 5 % of each other — the fastest will do, unless an earlier walk of the context has
        // measured something clearly better (then this is a neighbourhood of one kind, not a box without kinds: one buffer set of cfg 5 settled for a 6.0 ten
        // candidates into such a stretch while its context had kept a 6.8; not for arenas either: their walk is cheap and every lease seen had a clear pairing somewhere)
        if (!arena && i >= 9 && b < 1.05 * worst() && b >= kPlaceNearBest * seen()) return PlaceVerdict::settle;
        return PlaceVerdict::go_on;
    }
};

}  // namespace hrx
