// hrx_kernel.hpp — host-visible launch interface of the HIP kernels (hrx_kernel.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

#include "hrx_defs.hpp"

namespace hrx {

struct WitnessArgs {
    const uint8_t *chars;
    uint64_t stride;
    const uint32_t *lens;
    uint32_t B, M;
    uint32_t rec_pitch, msk_pitch;  // rows between consecutive strings in records / masked (>= M), string-major layout
    uint32_t layout;                // 0 string-major [B][pitch][D] / [B][pitch]; 1 position-major [M/4][D][B][4] / [M/8][B][8]
    uint32_t *records;
    uint16_t *masked;
    uint64_t *status;
    const uint32_t *table_image;  // device copy of DefsSet::table_image
    uint32_t table_bytes;
    const uint64_t *wide_image;   // device copy of DefsSet::wide_image (same byte size as table_image), or NULL
    const uint16_t *half_image;   // device copy of DefsSet::half_image (the exact LDS image, half_bytes long), or NULL
    uint32_t half_bytes;
    uint32_t n_groups;            // ceil(B / gs), set by plan_witness_launch
    uint32_t gs;                  // strings per wave (64, 32 or 16), set by plan_witness_launch
    uint32_t D;
    uint32_t debug;               // HRX_DEBUG_FLAGS (profiling ablations only): 1 skip record stores, 2 skip masked stores
    unsigned long long *stamps;   // profiling only (tools/kbench): per wave and tile 4 s_memtime stamps; NULL in the product
    DefConsts dc[3];
};

struct LaunchInfo {
    int split;         // 2: loader/walker kernel for the position-major layout (witness_pm_kernel),
                       // 1: walker/storer kernel (witness_split_kernel), 0: one-wave-does-all kernel (witness_kernel)
    int waves_per_wg;  // split: 2 * pairs
    int nslots;        // split: ring slots per walker/storer pair
    int gtab;          // 1: fused table read from global memory (too large for LDS)
    int wide;          // 1: position-major kernel on the WIDE table (hrx_lane.h)
    int half;          // 1: position-major kernel on the HALF table (hrx_lane.h)
    int grid;
    size_t lds_bytes;
};

// LDS bytes one wave stages per 64-string x 64-row tile
constexpr size_t wave_stage_bytes(int D, unsigned gs) { return ((size_t)gs + 1) * ((256 * (size_t)D + 16) + 80 + 8); }
constexpr size_t kLdsLimit = 160 * 1024;

// Picks the launch geometry for `a` on a device with `num_cus` CUs; returns false if nothing fits.
bool plan_witness_launch(WitnessArgs &a, int num_cus, LaunchInfo &out);
hipError_t launch_witness(const WitnessArgs &a, const LaunchInfo &li, hipStream_t stream);

// states-in entry points (lib.rs:825-888): tags[d*n+i] = pair_tag(states[d][i], states[d][i+1])
hipError_t launch_pair_tags(const uint64_t *states, size_t n, uint32_t D, const uint16_t *const *pair_tags,
                            const uint32_t *n_states, uint16_t *tags, hipStream_t stream);

// derive_is_start_end (lib.rs:847-888) for caller-supplied states AND substr ids:
// flags[d*n+i] bit0 = is_start[d][i], bit1 = is_end[d][i+1]
struct EndpointArgs {
    const uint64_t *states;
    const uint64_t *substr_ids;
    uint64_t n;
    uint32_t D;
    const uint8_t *member[3];
    uint32_t n_states[3], n_substrs[3], id_offset[3];
    uint8_t *flags;
};
hipError_t launch_endpoint_flags(const EndpointArgs &a, hipStream_t stream);

// compact witness rows of strings [b_begin, b_begin + b_count) -> bn256::Fr cells, [col][string][row][4]
struct FrArgs {
    const uint8_t *chars;
    uint64_t stride;
    const uint32_t *lens;
    const uint32_t *records;
    const uint16_t *masked;
    uint32_t B, M, D, layout, rec_pitch, msk_pitch;
    uint32_t b_begin, b_count;
    uint32_t canonical;   // 1: plain integers instead of Montgomery form
    uint64_t *cells;
};
hipError_t launch_fr_columns(const FrArgs &a, hipStream_t stream);

}  // namespace hrx
