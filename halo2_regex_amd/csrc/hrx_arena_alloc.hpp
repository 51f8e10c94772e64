// hrx_arena_alloc.hpp — the offsets inside one placement arena (hrx_api.cpp hrx_place_arena): first fit over a sorted list of free ranges, freed ranges merge
// with their neighbours.  Pure bookkeeping (no device call), so that tests/host_cpp/test_arena_alloc.cpp can exercise it on a host: a prover that allocates and
// frees its output buffers per batch must not wear a measured arena pair out — the bump pointer of round 3 only started over when EVERY sub-buffer was gone,
// and a steady alloc / free churn walked for a new pair every eight allocations.
#pragma once
#include <cstddef>
#include <map>

namespace hrx {

class ArenaRanges {
public:
    ArenaRanges() = default;
    explicit ArenaRanges(size_t bytes) { reset(bytes); }
    void reset(size_t bytes) { bytes_ = bytes; free_.clear(); if (bytes) free_[0] = bytes; used_.clear(); }
    size_t bytes() const { return bytes_; }
    size_t live() const { return used_.size(); }
    // can `need` bytes be taken right now?
    bool fits(size_t need) const {
        for (const auto &f : free_) if (f.second >= need) return true;
        return false;
    }
    // first fit; returns the offset, or (size_t)-1
    size_t take(size_t need) {
        if (need == 0) return (size_t)-1;
        for (auto it = free_.begin(); it != free_.end(); ++it) {
            if (it->second < need) continue;
            const size_t off = it->first, rest = it->second - need;
            free_.erase(it);
            if (rest) free_[off + need] = rest;
            used_[off] = need;
            return off;
        }
        return (size_t)-1;
    }
    // gives a range taken before back; false if `off` is not the start of a live range
    bool give(size_t off) {
        auto u = used_.find(off);
        if (u == used_.end()) return false;
        size_t lo = off, len = u->second;
        used_.erase(u);
        auto next = free_.lower_bound(lo);
        if (next != free_.end() && lo + len == next->first) { len += next->second; next = free_.erase(next); }
        if (next != free_.begin()) {
            auto prev = std::prev(next);
            if (prev->first + prev->second == lo) { lo = prev->first; len += prev->second; free_.erase(prev); }
        }
        free_[lo] = len;
        return true;
    }
    size_t largest_free() const { size_t m = 0; for (const auto &f : free_) if (f.second > m) m = f.second; return m; }

private:
    size_t bytes_ = 0;
    std::map<size_t, size_t> free_;   // offset -> length, disjoint, never adjacent
    std::map<size_t, size_t> used_;   // offset -> length of the live ranges
};

}  // namespace hrx
