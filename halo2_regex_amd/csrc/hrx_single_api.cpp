// hrx_single_api.cpp — the reference-shaped single-string entry points (src/lib.rs:316-318, 311-773): hrx_derive_states / _substr_ids / _is_start_end, hrx_match_substrs.
// One string is a batch of one: the native host walk (hrx_host_walk.cpp) unless HRX_DEBUG_FLAGS forces the device path.
#include "hrx_ctx.hpp"
#include "hrx_host_walk.hpp"
#include "hrx_lane.h"

using namespace hrx;

extern "C" {

/* ------------------------------ single-string entry points ------------------------------ */

static int status_to_error(uint64_t sw) {
    switch (sw & 0xff) {
        case kStatusOk: return HRX_OK;
        case kStatusInvalidTransition: {
            char buf[96];
            // the reference's panic text, lib.rs:817
            std::snprintf(buf, sizeof buf, "The transition from %u by %u is invalid!", (unsigned)((sw >> 24) & 0xffff),
                          (unsigned)((sw >> 16) & 0xff));
            return fail(HRX_ERR_INVALID_TRANSITION, buf);
        }
        case kStatusFlagOverlap:
            return fail(HRX_ERR_OUT_OF_CONTRACT, "two regex defs raise a start/end flag on row " + std::to_string(sw >> 40));
        default: return fail(HRX_ERR_OUT_OF_CONTRACT, "input longer than max_chars_size");
    }
}

// One string with M rows -> host copies of the compact outputs.  A single string is the host walk's case (one GPU lane
// needs ~50 ns per row, a host core ~3); the batch kernel serves it only when HRX_DEBUG_FLAGS says so (the GPU tests).
static int run_one(hrx_ctx *ctx, const uint8_t *characters, size_t n, size_t M, std::vector<uint32_t> &rec,
                   std::vector<uint16_t> &msk, uint64_t &sw) {
    const size_t D = ctx->s.defs.size();
    rec.assign(M * D, 0);
    msk.assign(M, 0);
    if (int rc = check_host_shape(1, M)) return rc;
    if (ctx->device == HRX_DEVICE_NONE || !(ctx->debug & kDbgNoHost)) {
        sw = host_witness_one(ctx->s, characters, n, M, rec.data(), msk.data());
        return HRX_OK;
    }
    const uint32_t len = (uint32_t)n;
    std::vector<uint8_t> tmp((n + 15) & ~(size_t)15, 0);
    if (n) std::memcpy(tmp.data(), characters, n);
    if (tmp.empty()) tmp.resize(16, 0);
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard guard;
    HIP_TRY(guard.set(ctx->device));
    return batch_host_locked(ctx, tmp.data(), tmp.size(), &len, 1, M, rec.data(), msk.data(), &sw);
}

static bool single_on_device(const hrx_ctx *ctx) { return ctx->device != HRX_DEVICE_NONE && (ctx->debug & kDbgNoHost); }

int hrx_derive_states(hrx_ctx *ctx, const uint8_t *characters, size_t n, uint64_t *states) {
    if (!ctx || (!characters && n) || !states) return fail(HRX_ERR_ARG, "NULL argument");
    const size_t D = ctx->s.defs.size();
    if (!single_on_device(ctx)) {
        uint32_t bs = 0, bc = 0;
        if (!host_derive_states(ctx->s, characters, n, states, bs, bc)) return status_to_error(status_invalid(0, 0, bs, bc));
        return HRX_OK;
    }
    const size_t M = n + 1;  // row n holds states[d][n] (lib.rs:406-411)
    std::vector<uint32_t> rec;
    std::vector<uint16_t> msk;
    uint64_t sw = 0;
    if (int rc = run_one(ctx, characters, n, M, rec, msk, sw)) return rc;
    if ((sw & 0xff) == kStatusInvalidTransition) return status_to_error(sw);   // (records of such a string are unspecified)
    for (size_t d = 0; d < D; ++d)
        for (size_t i = 0; i <= n; ++i) states[d * (n + 1) + i] = rec[i * D + d] & 0xffffu;
    return HRX_OK;
}

static int pair_tags_any(hrx_ctx *ctx, const uint64_t *states, size_t n, std::vector<uint16_t> &tags) {
    const size_t D = ctx->s.defs.size();
    tags.assign(n * D, 0);
    if (n == 0) return HRX_OK;
    if (!single_on_device(ctx)) {
        host_pair_tags(ctx->s, states, n, tags.data());
        return HRX_OK;
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard guard;
    HIP_TRY(guard.set(ctx->device));
    HIP_TRY(ctx->states.reserve(8 * D * (n + 1)));
    HIP_TRY(ctx->tags.reserve(2 * D * n));
    HIP_TRY(hipMemcpyAsync(ctx->states.p, states, 8 * D * (n + 1), hipMemcpyHostToDevice, ctx->stream));
    std::vector<uint32_t> ns(D);
    std::vector<const uint16_t *> pt(D);
    for (size_t d = 0; d < D; ++d) { ns[d] = (uint32_t)ctx->s.defs[d].allstr.largest_state_val + 1; pt[d] = ctx->d_pair[d]; }
    HIP_TRY(launch_pair_tags((const uint64_t *)ctx->states.p, n, (uint32_t)D, pt.data(), ns.data(), (uint16_t *)ctx->tags.p, ctx->stream));
    HIP_TRY(hipMemcpyAsync(tags.data(), ctx->tags.p, 2 * D * n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return HRX_OK;
}

int hrx_derive_substr_ids(hrx_ctx *ctx, const uint64_t *states, size_t n, uint64_t *substr_ids) {
    if (!ctx || !states || (!substr_ids && n)) return fail(HRX_ERR_ARG, "NULL argument");
    std::vector<uint16_t> tags;
    if (int rc = pair_tags_any(ctx, states, n, tags)) return rc;
    for (size_t i = 0; i < tags.size(); ++i) substr_ids[i] = tags[i] & 0xffu;
    return HRX_OK;
}

int hrx_derive_is_start_end(hrx_ctx *ctx, const uint64_t *states, const uint64_t *substr_ids, size_t n,
                            uint8_t *is_start, uint8_t *is_end) {
    if (!ctx || !states || !is_start || !is_end || (!substr_ids && n)) return fail(HRX_ERR_ARG, "NULL argument");
    const size_t D = ctx->s.defs.size();
    std::vector<uint8_t> flags(n * D, 0);
    if (n && !single_on_device(ctx)) {
        host_endpoint_flags(ctx->s, states, substr_ids, n, flags.data());
    } else if (n) {
        std::lock_guard<std::mutex> lk(ctx->mu);
        DeviceGuard guard;
        HIP_TRY(guard.set(ctx->device));
        HIP_TRY(ctx->states.reserve(8 * D * (n + 1)));
        HIP_TRY(ctx->tags.reserve(8 * D * n + D * n));
        uint64_t *d_sids = (uint64_t *)ctx->tags.p;
        uint8_t *d_flags = (uint8_t *)ctx->tags.p + 8 * D * n;
        HIP_TRY(hipMemcpyAsync(ctx->states.p, states, 8 * D * (n + 1), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(d_sids, substr_ids, 8 * D * n, hipMemcpyHostToDevice, ctx->stream));
        EndpointArgs a{};
        a.states = (const uint64_t *)ctx->states.p; a.substr_ids = d_sids; a.n = n; a.D = (uint32_t)D; a.flags = d_flags;
        std::vector<const uint8_t *> member(D);
        std::vector<uint32_t> dims(3 * D);
        for (size_t d = 0; d < D; ++d) {
            member[d] = ctx->d_member[d];
            dims[3 * d] = (uint32_t)ctx->s.defs[d].allstr.largest_state_val + 1;
            dims[3 * d + 1] = (uint32_t)ctx->s.defs[d].substrs.size();
            dims[3 * d + 2] = ctx->s.consts[d].substr_id_offset;
        }
        HIP_TRY(launch_endpoint_flags(a, member.data(), dims.data(), ctx->stream));
        HIP_TRY(hipMemcpyAsync(flags.data(), d_flags, D * n, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    for (size_t d = 0; d < D; ++d) {
        for (size_t i = 0; i < n; ++i) {
            is_start[d * (n + 1) + i] = flags[d * n + i] & 1;
            is_end[d * (n + 1) + i + 1] = (flags[d * n + i] >> 1) & 1;
        }
        is_start[d * (n + 1) + n] = 0;  // lib.rs:869
        is_end[d * (n + 1)] = 0;        // lib.rs:882
    }
    return HRX_OK;
}

int hrx_match_substrs(hrx_ctx *ctx, const uint8_t *characters, size_t n, size_t M, uint64_t *enable, uint64_t *character,
                      uint64_t *state, uint64_t *substr_id, uint64_t *start_enable, uint64_t *end_enable,
                      uint64_t *masked_char, uint64_t *masked_substr_id, uint64_t *status) {
    if (!ctx || (!characters && n)) return fail(HRX_ERR_ARG, "NULL argument");
    if (n > M) return fail(HRX_ERR_OUT_OF_CONTRACT, "input longer than max_chars_size");
    const size_t D = ctx->s.defs.size();
    std::vector<uint32_t> rec;
    std::vector<uint16_t> msk;
    uint64_t sw = 0;
    if (int rc = run_one(ctx, characters, n, M, rec, msk, sw)) return rc;
    if (status) *status = sw;
    if (int rc = status_to_error(sw)) return rc;
    for (size_t r = 0; r < M; ++r) {
        if (enable) enable[r] = r < n ? 1 : 0;                          // lib.rs:339-348
        if (character) character[r] = r < n ? characters[r] : 0;
        for (size_t d = 0; d < D; ++d) {
            const uint32_t w = rec[r * D + d];
            if (state) state[d * M + r] = w & 0xffffu;
            if (substr_id) substr_id[d * M + r] = (w >> 16) & 0xffu;
            if (start_enable) start_enable[d * M + r] = (w >> 24) & 1u;
            if (end_enable) end_enable[d * M + r] = (w >> 25) & 1u;
        }
        if (masked_char) masked_char[r] = msk[r] & 0xffu;
        if (masked_substr_id) masked_substr_id[r] = msk[r] >> 8;
    }
    return HRX_OK;
}

}  // extern "C"
