// The placement walk's stopping rule (csrc/hrx_place_rule.hpp) replayed on candidate sequences the MI355X leases of round 4 recorded (GB/s of the two-stream probe;
// profiles/r04_probes/cfg5_batch_sweep.txt, profiles/r04_config_sweep_new_rule/, profiles/r04_probes/headline_leases.txt).  No device: the rule is pure arithmetic.
#include <cstdio>
#include <cstdlib>
#include <initializer_list>

#include "../../halo2_regex_amd/csrc/hrx_place_rule.hpp"

using hrx::PlaceVerdict;
using hrx::PlaceWalk;

static int failures = 0;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #c); ++failures; } } while (0)

struct Outcome { int steps; PlaceVerdict verdict; double kept; };
// feeds the candidates in order (elapsed time: ms_per_step each) until the rule stops the walk or they run out
static Outcome run(PlaceWalk w, std::initializer_list<double> cands, double ms_per_step = 2.0) {
    int i = 0;
    for (double c : cands) {
        if (!w.may_take_another()) return {i, PlaceVerdict::settle, w.best()};
        w.rates.push_back(c);
        ++i;
        const PlaceVerdict v = w.decide(ms_per_step * i);
        if (v != PlaceVerdict::go_on) return {i, v, w.best()};
    }
    return {i, PlaceVerdict::go_on, w.best()};
}

int main() {
    {   // cfg 5, 393216 x 4096, buffer set 0: a 4.7 first, then a 6.1 — round 3's rule (10 % above the slowest of two) stopped here and the launch ran at 0.55
        PlaceWalk w; w.ref_rate = 4623;
        Outcome o = run(w, {4730, 6118});
        CHECK(o.verdict == PlaceVerdict::go_on);
        // ... the walk goes on through the middle kind until a clear one comes
        o = run(w, {4730, 6118, 5800, 5900, 6050, 6922, 5700, 5750, 5800});
        CHECK(o.verdict == PlaceVerdict::accept && o.steps == 8 && o.kept == 6922);      // (a context's first direct walk looks at eight candidates: round 5)
    }
    {   // a clear candidate first still needs its company (the median must be a colliding one): eight candidates in a context's first direct walk, four later and in arena walks
        PlaceWalk w; w.ref_rate = 5412;
        Outcome o = run(w, {7205, 4020, 5600, 5750, 5800, 5700, 5650, 5600, 5900});
        CHECK(o.verdict == PlaceVerdict::accept && o.steps == 8 && o.kept == 7205);
        PlaceWalk a; a.ref_rate = 5412; a.arena = true;
        o = run(a, {7205, 4020, 5600, 5750, 5800});
        CHECK(o.verdict == PlaceVerdict::accept && o.steps == 4 && o.kept == 7205);
    }
    {   // two clear ones among the first candidates do not hide each other (lower median)
        PlaceWalk w; w.ref_rate = 5400; w.arena = true;
        Outcome o = run(w, {7000, 5800, 7050, 5900});
        CHECK(o.verdict == PlaceVerdict::accept && o.steps == 4 && o.kept == 7050);
    }
    {   // round 5, lease b, cfg 3's first buffer set: a hard-kind reference, and after four candidates a middle-kind 5.9 clears reference and median by 10 % — four more candidates
        // turn up what the lease has (the same lease's other process kept 6.8 / 7.0)
        PlaceWalk w; w.ref_rate = 4660;
        Outcome o = run(w, {4764, 4700, 4810, 5899});
        CHECK(o.verdict == PlaceVerdict::go_on);
        o = run(w, {4764, 4700, 4810, 5899, 5850, 6809, 4790, 5900});
        CHECK(o.verdict == PlaceVerdict::accept && o.steps == 8 && o.kept == 6809);
    }
    {   // a later buffer set of the same context: as good as what an earlier walk kept ends the walk at its second candidate ...
        PlaceWalk w; w.ref_rate = 5412; w.seen_before = 7239;
        Outcome o = run(w, {5653, 7170, 5000, 5000});
        CHECK(o.verdict == PlaceVerdict::accept && o.steps == 2 && o.kept == 7170);
        // ... and a middle-kind candidate that clears reference and median is NOT enough while the context knows better (within 4 %)
        o = run(w, {4772, 5700, 5650, 6600, 5600, 5500});
        CHECK(o.verdict == PlaceVerdict::go_on);
        o = run(w, {4772, 5700, 5650, 6600, 5600, 7068});
        CHECK(o.verdict == PlaceVerdict::accept && o.steps == 6 && o.kept == 7068);
    }
    {   // neighbourhood of mostly clear candidates: the median rule cannot fire (7.0 < 1.1 x 6.9) — the time bound ends the walk past its eighth candidate
        PlaceWalk w; w.ref_rate = 5987;
        Outcome o = run(w, {5809, 6900, 6950, 7000, 6900, 7010, 6950, 7314, 6900, 6900, 6900, 6900}, 40.0);
        CHECK(o.verdict == PlaceVerdict::settle && o.steps == 8 && o.kept == 7314);
        CHECK(run(w, {5809, 6900, 6950, 7000, 6900, 7010, 6950, 7314}, 2.0).verdict == PlaceVerdict::go_on);
    }
    {   // no kinds at all (direct walk): ten candidates within 5 % of each other
        PlaceWalk w; w.ref_rate = 6000;
        Outcome o = run(w, {6100, 6150, 6120, 6200, 6180, 6110, 6130, 6160, 6190, 6170, 6100});
        CHECK(o.verdict == PlaceVerdict::settle && o.steps == 10);
    }
    {   // ... but not while the context knows of a better kind: a stretch of ten middle-kind candidates is a neighbourhood, not a box without kinds (cfg 5, 65536 x 4096, set 1 of 8)
        PlaceWalk w; w.ref_rate = 5398; w.seen_before = 6840;
        Outcome o = run(w, {5900, 5950, 5920, 5995, 5890, 5910, 5940, 5960, 5930, 5905, 5915, 7010});
        CHECK(o.verdict == PlaceVerdict::accept && o.steps == 12 && o.kept == 7010);
    }
    {   // the arena walk of the lease behind profiles/r04_config_sweep_new_rule/: 24 candidates between 5.6 and 6.07 against a reference of 5.9 — nothing clear of the
        // reference, so the walk may pass the soft cap ...
        PlaceWalk w; w.ref_rate = 5896; w.arena = true;
        for (int i = 0; i < 24; ++i) { w.rates.push_back(5600 + 20.0 * (i % 24)); CHECK(w.decide(2.0 * (i + 1)) == PlaceVerdict::go_on); }
        CHECK(w.may_take_another());
        // ... until a clear pairing turns up
        w.rates.push_back(7100);
        CHECK(w.decide(60.0) == PlaceVerdict::accept);
    }
    {   // an arena walk with a mid-high neighbourhood (lease 2 of headline_leases.txt): stops AT the soft cap with its clear best, not at the hard cap
        PlaceWalk w; w.ref_rate = 5871; w.arena = true;
        for (int i = 0; i < 24; ++i) { CHECK(w.may_take_another()); w.rates.push_back(i == 3 ? 7275 : 6800 + 5.0 * i); CHECK(w.decide(2.0 * (i + 1)) == PlaceVerdict::go_on); }
        CHECK(!w.may_take_another());
        CHECK(w.clear_of_reference());
    }
    {   // a box whose 2-GiB allocations take 66 ms each (seen): the soft time bound must not end an arena walk that holds nothing clear of the reference ...
        PlaceWalk w; w.ref_rate = 5797; w.arena = true;
        Outcome o = run(w, {6032, 6178, 5900, 5950, 6000, 6100, 5990, 6050, 6010, 6120, 5980, 6040, 6060, 6020, 6090, 6030, 6070, 7150}, 66.0);
        CHECK(o.verdict == PlaceVerdict::accept && o.steps == 18 && o.kept == 7150);
        // ... only the hard bound does (8 s), and a direct walk's after 2 s
        o = run(w, {6032, 6178, 5900, 5950, 6000, 6100, 5990, 6050, 6010, 6120}, 1000.0);
        CHECK(o.verdict == PlaceVerdict::settle && o.steps == 9);
        PlaceWalk d; d.ref_rate = 5797;
        o = run(d, {6032, 6178, 5900, 5950, 6000, 6100, 5990, 6050, 6010, 6120}, 300.0);
        CHECK(o.verdict == PlaceVerdict::settle && o.steps == 7);
    }
    {   // failed probes (rate 0) neither count as candidates of a kind nor crash the median
        PlaceWalk w; w.ref_rate = 5000;
        Outcome o = run(w, {0, 0, 0, 0});
        CHECK(o.verdict == PlaceVerdict::go_on && o.kept == 0);
        PlaceWalk z;   // no reference measured: never accept on arithmetic alone
        CHECK(run(z, {7000, 5000, 5000, 5000}).verdict == PlaceVerdict::go_on);
    }
    // ---- fresh synthetic sequences (round 5): random mixes of the three kinds of candidates the leases showed — pairings that collide hard (4.6-4.8 TB/s on the
    // probe), the middle kind (5.6-6.1) and clear ones (6.9-7.3) — drawn with their observed frequencies (clear: about one in eight), against a same-block reference
    // of the hard or the middle kind, direct and arena walks, fast and slow allocations.  Not outcomes the rule was tuned on: invariants it must keep on any of them.
    {
        unsigned long long st = 0x9e3779b97f4a7c15ull;
        auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (double)((st >> 33) & 0xffffff) / (double)0x1000000; };
        auto draw = [&](double p_clear, double p_hard) {
            const double u = rnd();
            if (u < p_clear) return 6900.0 + 400.0 * rnd();
            if (u < p_clear + p_hard) return 4600.0 + 200.0 * rnd();
            return 5600.0 + 500.0 * rnd();
        };
        int walks = 0, accepted = 0, accepted_clear = 0, ended_without_clear_although_seen = 0, total_steps = 0;
        for (int seed = 0; seed < 4000; ++seed) {
            PlaceWalk w;
            w.arena = (seed & 1) != 0;
            w.ref_rate = (seed & 2) ? 4600.0 + 200.0 * rnd() : 5700.0 + 300.0 * rnd();
            if (seed % 5 == 0) w.seen_before = 6900.0 + 400.0 * rnd();          // a later buffer set: an earlier walk kept a clear pairing
            const double p_clear = (seed % 7 == 0) ? 0.0 : 0.125, p_hard = 0.1 + 0.3 * rnd();
            const double ms_per_step = (seed % 11 == 0) ? 66.0 : 2.0 + 30.0 * rnd();      // (2-GiB hipMallocs took 2 ms on most leases, 66 on one)
            const int cap = w.arena ? hrx::kPlaceArenaHardSteps : 48;
            PlaceVerdict v = PlaceVerdict::go_on;
            int i = 0;
            bool saw_clear = false;
            for (; i < cap; ++i) {
                if (!w.may_take_another()) { v = PlaceVerdict::settle; break; }
                const double r = draw(p_clear, p_hard);
                saw_clear = saw_clear || r >= 6900.0;
                w.rates.push_back(r);
                v = w.decide(ms_per_step * (i + 1));
                if (v != PlaceVerdict::go_on) { ++i; break; }
            }
            ++walks; total_steps += i;
            const double hard_ms = w.arena ? hrx::kPlaceArenaHardMs : hrx::kPlaceHardMs;
            CHECK(ms_per_step * (i - 1) <= hard_ms);                                  // no walk outlives the hard time bound by more than one candidate
            CHECK(w.best() >= w.worst());
            if (v == PlaceVerdict::accept) {
                ++accepted;
                CHECK(w.best() >= hrx::kPlaceMargin * w.ref_rate);                    // an accepted pairing is clear of the same-block reference ...
                CHECK(w.seen_before > 0 || (int)w.rates.size() >= hrx::kPlaceMinCandidates);      // ... never on fewer than four candidates unless an earlier walk vouches
                CHECK(w.best() >= hrx::kPlaceNearBest * w.seen() || w.best() >= hrx::kPlaceAsSeen * w.seen_before);
                if (w.best() >= 6900.0) ++accepted_clear;
                // a middle-kind candidate is only ever accepted against a HARD-kind reference and median (there it is the better class: + 20 %)
                if (w.best() < 6900.0) CHECK(w.ref_rate < 5000.0 && w.median() < 0.91 * w.best());
            }
            // a walk that saw a clear candidate keeps it (the kept one is the fastest measured, whatever ended the walk)
            if (saw_clear) CHECK(w.best() >= 6900.0);
            if (p_clear == 0.0) CHECK(w.best() < 6900.0);
            // with a middle-kind reference nothing but a clear candidate may be ACCEPTED
            if (v == PlaceVerdict::accept && w.ref_rate >= 5700.0) CHECK(w.best() >= 6900.0);
            if (v != PlaceVerdict::accept && saw_clear && w.ref_rate >= 5700.0 && ms_per_step * i < 250.0) ++ended_without_clear_although_seen;
        }
        // the rule finds the clear kind when it is there and the walk is cheap: a clear candidate in a cheap walk is accepted, not walked past
        CHECK(ended_without_clear_although_seen == 0);
        CHECK(accepted > walks / 2 && accepted_clear > accepted * 3 / 4);
        CHECK(total_steps / walks < 24);                                              // (one clear pairing in eight: ~8-12 candidates on average)
        std::printf("place rule, synthetic mixes: %d walks, %d accepted (%d of the clear kind), %.1f candidates on average\n", walks, accepted, accepted_clear,
                    (double)total_steps / walks);
    }
    if (failures) return 1;
    std::puts("place rule: ok");
    return 0;
}
