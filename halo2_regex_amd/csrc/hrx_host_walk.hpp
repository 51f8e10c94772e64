// hrx_host_walk.hpp — native small-batch host path (hrx_host_walk.cpp): the lane algorithm of hrx_lane.h run on a host core.
#pragma once
#include <cstddef>
#include <cstdint>

#include "hrx_defs.hpp"

namespace hrx {

// one string -> compact rows (records [M][D], masked [M]); returns the status word of include/hrx.h
uint64_t host_witness_one(const DefsSet &s, const uint8_t *chars, size_t n, size_t M, uint32_t *records, uint16_t *masked);
// string-major batch, `threads` host threads (contiguous slices of the batch)
void host_witness_batch(const DefsSet &s, const uint8_t *chars, size_t stride, const uint32_t *lens, size_t B, size_t M,
                        uint32_t *records, uint16_t *masked, uint64_t *status, int threads);
bool host_derive_states(const DefsSet &s, const uint8_t *chars, size_t n, uint64_t *states, uint32_t &bad_state, uint32_t &bad_char);
void host_pair_tags(const DefsSet &s, const uint64_t *states, size_t n, uint16_t *tags);
void host_endpoint_flags(const DefsSet &s, const uint64_t *states, const uint64_t *substr_ids, size_t n, uint8_t *flags);

}  // namespace hrx
