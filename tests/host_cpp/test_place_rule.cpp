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
        o = run(w, {4730, 6118, 5800, 5900, 6050, 6922, 5700});
        CHECK(o.verdict == PlaceVerdict::accept && o.steps == 6 && o.kept == 6922);
    }
    {   // a clear candidate first still needs four candidates (the median must be a colliding one), then wins
        PlaceWalk w; w.ref_rate = 5412;
        Outcome o = run(w, {7205, 4020, 5600, 5750, 5800});
        CHECK(o.verdict == PlaceVerdict::accept && o.steps == 4 && o.kept == 7205);
    }
    {   // two clear ones among the first four do not hide each other (lower median)
        PlaceWalk w; w.ref_rate = 5400;
        Outcome o = run(w, {7000, 5800, 7050, 5900});
        CHECK(o.verdict == PlaceVerdict::accept && o.steps == 4 && o.kept == 7050);
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
    if (failures) return 1;
    std::puts("place rule: ok");
    return 0;
}
