// hrx_fill.cpp — SURVEY §8 f3, the executable half: from the compact witness records to the values the reference's fill consumes.  Host only, no context, re-entrant.
//
// RegexVerifyConfig::match_substrs (src/lib.rs:311-773) begins with the three derive_* calls (:316-318); everything below them reads only their four results,
// `characters` and `self`.  A batch-aware fill therefore runs the body below :318 per circuit, fed from the records of the circuit's string:
//   hrx_witness_of_string    -> exactly the Vecs of lib.rs:316-318 (what bindings/rust/hrx.rs WitnessOf holds);
//   hrx_witness_columns_host -> the integer content of every advice column the loops at lib.rs:339-348, 419-519 assign and of the two result columns (:752-771), column-major
//                               over the circuits of the batch — the host-side, plain-integer twin of hrx_fr_columns_device (same column order).
#include <algorithm>
#include <cstring>
#include <string>

#include "../../include/hrx.h"
#include "hrx_error.hpp"

namespace {
inline int bad(const char *msg) { return hrx::set_last_error(HRX_ERR_ARG, msg); }
}  // namespace

extern "C" {

int hrx_witness_of_string(const uint32_t *records, size_t D, size_t n, size_t M, uint64_t *states, size_t *substr_ids, uint8_t *is_start, uint8_t *is_end) {
    if (!records || !states || !substr_ids || !is_start || !is_end) return bad("hrx_witness_of_string: NULL buffer");
    if (D == 0 || D > HRX_MAX_DEFS || M == 0 || n > M) return bad("hrx_witness_of_string: shape out of range (1 <= D <= HRX_MAX_DEFS, n <= M)");
    for (size_t d = 0; d < D; ++d) {
        uint64_t *s = states + d * (n + 1);
        size_t *id = substr_ids + d * n;
        uint8_t *st = is_start + d * (n + 1), *en = is_end + d * (n + 1);
        en[0] = 0;                                             // is_ends[d][0] = false: lib.rs:881
        for (size_t i = 0; i < n; ++i) {
            const uint32_t r = records[i * D + d];
            s[i] = r & 0xffffu;                                // states[d][i]: the state BEFORE character i (lib.rs:807-819)
            id[i] = (r >> 16) & 0xffu;                         // substr_ids[d][i]: lib.rs:829-842
            st[i] = (uint8_t)((r >> 24) & 1u);                 // enable[i] = 1 for i < n, so start_enable[i] = is_starts[d][i] (lib.rs:482-493)
            en[i + 1] = (uint8_t)((r >> 25) & 1u);             // ... and end_enable[i] = is_ends[d][i + 1] (lib.rs:501-519)
        }
        // row n holds the state after the last character (lib.rs:404-411).  n == M: that row does not exist — and the reference never assigns states[d][M] to any cell
        // (lib.rs:388-418 stop at max_chars_size), nor the last transition's end flag (lib.rs:501: the loop ends at max_chars_size - 2)
        s[n] = n < M ? (records[n * D + d] & 0xffffu) : 0;
        st[n] = 0;                                             // is_starts[d][n] = false: lib.rs:868
    }
    return HRX_OK;
}

size_t hrx_witness_num_columns(size_t D) { return 4 + 4 * D; }

int hrx_witness_columns_host(int layout, const uint8_t *chars, size_t stride, const uint32_t *lens, const uint32_t *records, size_t rec_pitch, const uint16_t *masked,
                             size_t msk_pitch, size_t B, size_t M, size_t D, size_t b_begin, size_t b_count, uint64_t *columns) {
    if (!chars || !lens || !records || !masked || !columns) return bad("hrx_witness_columns_host: NULL buffer");
    if (D == 0 || D > HRX_MAX_DEFS || M == 0 || b_begin > B || b_count > B - b_begin) return bad("hrx_witness_columns_host: shape out of range");
    if (layout != HRX_LAYOUT_STRING_MAJOR && layout != HRX_LAYOUT_POSITION_MAJOR && layout != (HRX_LAYOUT_POSITION_MAJOR | HRX_LAYOUT_INPUT_POSITION_MAJOR))
        return bad("hrx_witness_columns_host: unknown layout");
    const bool pm = (layout & HRX_LAYOUT_POSITION_MAJOR) != 0, in_pm = (layout & HRX_LAYOUT_INPUT_POSITION_MAJOR) != 0;
    if (!rec_pitch) rec_pitch = M;
    if (!msk_pitch) msk_pitch = M;
    if (!pm && (rec_pitch < M || msk_pitch < M)) return bad("hrx_witness_columns_host: pitches < M");
    const size_t q4 = (M + 3) / 4, q8 = (M + 7) / 8, col = b_count * M;
    for (size_t j = 0; j < b_count; ++j) {
        const size_t b = b_begin + j;
        const size_t k = b / HRX_PM_BLOCK, bl = b % HRX_PM_BLOCK, nb = std::min<size_t>(HRX_PM_BLOCK, B - k * HRX_PM_BLOCK);
        const size_t n = lens[b];
        if (n > M) return bad("hrx_witness_columns_host: a string longer than max_chars_size (its status word says so: no rows)");
        uint64_t *c0 = columns + j * M;
        auto rec_at = [&](size_t r, size_t d) -> uint32_t {
            return pm ? records[k * HRX_PM_BLOCK * q4 * D * 4 + ((r / 4 * D + d) * nb + bl) * 4 + r % 4] : records[(b * rec_pitch + r) * D + d];
        };
        auto msk_at = [&](size_t r) -> uint16_t { return pm ? masked[k * HRX_PM_BLOCK * q8 * 8 + (r / 8 * nb + bl) * 8 + r % 8] : masked[b * msk_pitch + r]; };
        auto chr_at = [&](size_t i) -> uint8_t { return in_pm ? chars[k * HRX_PM_BLOCK * stride + ((i / 16) * nb + bl) * 16 + i % 16] : chars[b * stride + i]; };
        for (size_t r = 0; r < M; ++r) {
            c0[0 * col + r] = r < n ? 1 : 0;                               // char_enable: lib.rs:339-348
            c0[1 * col + r] = r < n ? chr_at(r) : 0;                       // characters
            for (size_t d = 0; d < D; ++d) {
                const uint32_t x = rec_at(r, d);
                c0[(2 + 4 * d) * col + r] = x & 0xffffu;                   // states_array[d]: s[r], s[n], then largest + 1 (lib.rs:388-418)
                c0[(3 + 4 * d) * col + r] = (x >> 16) & 0xffu;             // substr_ids_array[d] (lib.rs:392-395, 459-471)
                c0[(4 + 4 * d) * col + r] = (x >> 24) & 1u;                // start_enable_array[d] (lib.rs:482-493)
                c0[(5 + 4 * d) * col + r] = r + 1 < M ? (x >> 25) & 1u : 0;  // end_enable_array[d]; row M - 1 is never assigned (lib.rs:501-519)
            }
            const uint16_t m = msk_at(r);
            c0[(2 + 4 * D) * col + r] = m & 0xffu;                         // masked_characters (lib.rs:752-756)
            c0[(3 + 4 * D) * col + r] = m >> 8;                            // all_substr_ids (lib.rs:757-761)
        }
    }
    return HRX_OK;
}

}  // extern "C"
