// hrx_api.cpp — the C ABI (include/hrx.h) over the host data model and the HIP kernels.
// Batches run on the gfx950 kernels; device-pointer entry points fail with HRX_ERR_HIP without a device.  The one
// host-side compute path is the native small-batch walk (hrx_host_walk.cpp): single strings and host-buffer batches
// below the context's threshold — a GPU launch cannot beat a host core on ~1000 rows.
#include <fstream>
#include <sstream>

#include "hrx_ctx.hpp"
#include "hrx_fr.h"
#include "hrx_host_walk.hpp"
#include "hrx_lane.h"

using namespace hrx;

static thread_local std::string g_err;
int hrx::set_last_error(int code, const std::string &msg) {   // hrx_error.hpp: every translation unit of the C ABI fails through here
    g_err = msg;
    return code;
}

namespace hrx {
uint32_t debug_flags_from_env() {
    const char *dbg = std::getenv("HRX_DEBUG_FLAGS");
    return dbg ? ((uint32_t)std::strtoul(dbg, nullptr, 0) & kDbgHonoured) : 0u;
}
}  // namespace hrx

static hipError_t upload_blob(const std::vector<uint8_t> &v, uint8_t *&d) {
    if (v.empty()) return hipSuccess;
    hipError_t e = hipMalloc((void **)&d, v.size());
    if (e == hipSuccess) e = hipMemcpy(d, v.data(), v.size(), hipMemcpyHostToDevice);
    return e;
}

static hipError_t upload_images(const DefsSet &s, uint32_t *&d_table, uint64_t *&d_wide, uint16_t *&d_half, uint8_t *&d_pairtab) {
    hipError_t e = hipMalloc((void **)&d_table, s.table_image.size() * 4);
    if (e == hipSuccess) e = hipMemcpy(d_table, s.table_image.data(), s.table_image.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && !s.wide_image.empty()) {
        e = hipMalloc((void **)&d_wide, s.wide_image.size() * 8);
        if (e == hipSuccess) e = hipMemcpy(d_wide, s.wide_image.data(), s.wide_image.size() * 8, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && !s.half_image.empty()) {
        e = hipMalloc((void **)&d_half, s.half_image.size() * 2);
        if (e == hipSuccess) e = hipMemcpy(d_half, s.half_image.data(), s.half_image.size() * 2, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && !s.pair.image.empty()) {
        e = hipMalloc((void **)&d_pairtab, s.pair.image.size());
        if (e == hipSuccess) e = hipMemcpy(d_pairtab, s.pair.image.data(), s.pair.image.size(), hipMemcpyHostToDevice);
    }
    return e;
}

extern "C" {

const char *hrx_last_error(void) { return g_err.c_str(); }

/* ------------------------------ data model ------------------------------ */

int hrx_defs_create(hrx_defs **out) {
    if (!out) return fail(HRX_ERR_ARG, "out is NULL");
    *out = new hrx_defs();
    return HRX_OK;
}

void hrx_defs_destroy(hrx_defs *defs) { delete defs; }

int hrx_defs_push_allstr_text(hrx_defs *defs, const char *text, size_t len) {
    if (!defs || (!text && len)) return fail(HRX_ERR_ARG, "NULL argument");
    if (defs->s.finalized) return fail(HRX_ERR_STATE, "defs already finalized");
    RegexDefs rd;
    const int rc = parse_allstr_text(text, len, rd.allstr);
    if (rc) return fail(HRX_ERR_PARSE, "allstr definition: cannot parse line " + std::to_string(-rc - 1));
    defs->s.defs.push_back(std::move(rd));
    return HRX_OK;
}

int hrx_defs_push_substr_text(hrx_defs *defs, const char *text, size_t len) {
    if (!defs || (!text && len)) return fail(HRX_ERR_ARG, "NULL argument");
    if (defs->s.finalized) return fail(HRX_ERR_STATE, "defs already finalized");
    if (defs->s.defs.empty()) return fail(HRX_ERR_STATE, "push an allstr definition first");
    SubstrRegexDef sd;
    const int rc = parse_substr_text(text, len, sd);
    if (rc) return fail(HRX_ERR_PARSE, "substr definition: cannot parse line " + std::to_string(-rc - 1));
    defs->s.defs.back().substrs.push_back(std::move(sd));
    return HRX_OK;
}

static int read_file(const char *path, std::string &out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return fail(HRX_ERR_IO, std::string("cannot open ") + (path ? path : "(null)"));  // File::open(..).unwrap()
    std::ostringstream ss;
    ss << f.rdbuf();
    out = ss.str();
    return HRX_OK;
}

int hrx_defs_push_allstr_file(hrx_defs *defs, const char *path) {
    std::string t;
    if (int rc = read_file(path, t)) return rc;
    return hrx_defs_push_allstr_text(defs, t.data(), t.size());
}

int hrx_defs_push_substr_file(hrx_defs *defs, const char *path) {
    std::string t;
    if (int rc = read_file(path, t)) return rc;
    return hrx_defs_push_substr_text(defs, t.data(), t.size());
}

int hrx_defs_push_allstr(hrx_defs *defs, uint64_t first_state_val, uint64_t accepted_state_val,
                         uint64_t largest_state_val, size_t n, const uint64_t *cur, const uint64_t *next,
                         const uint8_t *chr, const uint64_t *line_idx) {
    if (!defs || (n && (!cur || !next || !chr))) return fail(HRX_ERR_ARG, "NULL argument");
    if (defs->s.finalized) return fail(HRX_ERR_STATE, "defs already finalized");
    RegexDefs rd;
    rd.allstr.first_state_val = first_state_val;
    rd.allstr.accepted_state_val = accepted_state_val;
    rd.allstr.largest_state_val = largest_state_val;
    for (size_t i = 0; i < n; ++i)
        rd.allstr.state_lookup[{(uint64_t)chr[i], cur[i]}] = AllstrRegexDef::Val{line_idx ? line_idx[i] : (uint64_t)(i + 3), next[i]};
    defs->s.defs.push_back(std::move(rd));
    return HRX_OK;
}

int hrx_defs_push_substr(hrx_defs *defs, size_t n_pairs, const uint64_t *pair_cur, const uint64_t *pair_next,
                         size_t n_start, const uint64_t *start_states, size_t n_end, const uint64_t *end_states) {
    if (!defs || (n_pairs && (!pair_cur || !pair_next)) || (n_start && !start_states) || (n_end && !end_states))
        return fail(HRX_ERR_ARG, "NULL argument");
    if (defs->s.finalized) return fail(HRX_ERR_STATE, "defs already finalized");
    if (defs->s.defs.empty()) return fail(HRX_ERR_STATE, "push an allstr definition first");
    SubstrRegexDef sd;
    for (size_t i = 0; i < n_pairs; ++i) sd.valid_state_transitions.insert({pair_cur[i], pair_next[i]});
    if (n_start) sd.start_states.assign(start_states, start_states + n_start);
    if (n_end) sd.end_states.assign(end_states, end_states + n_end);
    defs->s.defs.back().substrs.push_back(std::move(sd));
    return HRX_OK;
}

int hrx_defs_finalize(hrx_defs *defs) {
    if (!defs) return fail(HRX_ERR_ARG, "NULL argument");
    std::string err;
    const int rc = finalize_defs(defs->s, err);
    return rc ? fail(rc, err) : HRX_OK;
}

size_t hrx_defs_num_defs(const hrx_defs *defs) { return defs ? defs->s.defs.size() : 0; }
size_t hrx_defs_num_substrs(const hrx_defs *defs, size_t d) { return defs && d < defs->s.defs.size() ? defs->s.defs[d].substrs.size() : 0; }
static const AllstrRegexDef *allstr_of(const hrx_defs *defs, size_t d) {   // NULL handle or def out of range -> NULL (accessors then return 0)
    return defs && d < defs->s.defs.size() ? &defs->s.defs[d].allstr : nullptr;
}
uint64_t hrx_defs_first_state(const hrx_defs *defs, size_t d) { const AllstrRegexDef *a = allstr_of(defs, d); return a ? a->first_state_val : 0; }
uint64_t hrx_defs_accepted_state(const hrx_defs *defs, size_t d) { const AllstrRegexDef *a = allstr_of(defs, d); return a ? a->accepted_state_val : 0; }
uint64_t hrx_defs_largest_state(const hrx_defs *defs, size_t d) { const AllstrRegexDef *a = allstr_of(defs, d); return a ? a->largest_state_val : 0; }
size_t hrx_defs_num_transitions(const hrx_defs *defs, size_t d) { const AllstrRegexDef *a = allstr_of(defs, d); return a ? a->state_lookup.size() : 0; }
uint64_t hrx_defs_substr_id_offset(const hrx_defs *defs, size_t d) {
    if (!defs) return 0;
    uint64_t off = 1;
    for (size_t i = 0; i < d && i < defs->s.defs.size(); ++i) off += defs->s.defs[i].substrs.size();
    return off;
}
size_t hrx_defs_table_bytes(const hrx_defs *defs) { return defs && defs->s.finalized ? defs->s.table_image.size() * 4 : 0; }

size_t hrx_table_transition_rows(const hrx_defs *defs, size_t d, uint64_t *rows4, size_t cap_rows) {
    if (!defs || !defs->s.finalized || d >= defs->s.defs.size()) return 0;
    return table_transition_rows(defs->s, d, rows4, cap_rows);
}

size_t hrx_table_endpoint_rows(const hrx_defs *defs, size_t d, uint64_t *rows3, size_t cap_rows) {
    if (!defs || !defs->s.finalized || d >= defs->s.defs.size()) return 0;
    return table_endpoint_rows(defs->s, d, rows3, cap_rows);
}

/* ------------------------------ device ------------------------------ */

int hrx_device_count(int *count) {
    if (!count) return fail(HRX_ERR_ARG, "NULL argument");
    *count = 0;
    HIP_TRY(hipGetDeviceCount(count));
    return HRX_OK;
}

void hrx_shard_range(size_t B, int world, int rank, size_t *begin, size_t *count) {
    if (world < 1) world = 1;
    const size_t per = (B + (size_t)world - 1) / (size_t)world;  // ceil(B / G) strings per device
    size_t b = per * (size_t)rank;
    if (b > B) b = B;
    size_t e = b + per;
    if (e > B) e = B;
    if (begin) *begin = b;
    if (count) *count = e - b;
}

static int ctx_create_from(const DefsSet &set, int device, hrx_ctx **out);

int hrx_ctx_create(const hrx_defs *defs, int device, hrx_ctx **out) {
    if (!defs || !out) return fail(HRX_ERR_ARG, "NULL argument");
    if (!defs->s.finalized) return fail(HRX_ERR_STATE, "call hrx_defs_finalize first");
    return ctx_create_from(defs->s, device, out);
}

int hrx_ctx_clone(const hrx_ctx *ctx, int device, hrx_ctx **out) {
    if (!ctx || !out) return fail(HRX_ERR_ARG, "NULL argument");
    hrx_ctx *c = nullptr;
    const int rc = ctx_create_from(ctx->s, device == HRX_DEVICE_SAME ? ctx->device : device, &c);
    if (rc != HRX_OK) return rc;
    c->host_threshold = ctx->host_threshold;      // the per-context switches travel with the clone
    c->tune = ctx->tune;
    c->place_dry = ctx->place_dry; c->host_route = ctx->host_route; c->host_threads = ctx->host_threads; c->host_pipeline = ctx->host_pipeline; c->host_chunk_mib = ctx->host_chunk_mib; c->host_trace = ctx->host_trace;
    c->place_enabled = ctx->place_enabled; c->place_max_bytes = ctx->place_max_bytes; c->place_max_ms = ctx->place_max_ms;
    *out = c;
    return HRX_OK;
}

static int ctx_create_from(const DefsSet &set, int device, hrx_ctx **out) {
    struct { const DefsSet &s; } defs_view{set};
    const auto *defs = &defs_view;
    if (device == HRX_DEVICE_NONE) {   // host-only context: single strings and host-buffer batches through the native host walk
        hrx_ctx *c = new hrx_ctx();
        c->s = defs->s;
        c->device = HRX_DEVICE_NONE;
        c->debug = debug_flags_from_env();
        *out = c;
        return HRX_OK;
    }
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return fail(HRX_ERR_ARG, "no such device");
    DeviceGuard guard;
    HIP_TRY(guard.set(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(HRX_ERR_HIP, std::string("kernels are built for gfx950 only; device is ") + prop.gcnArchName);
    hrx_ctx *c = new hrx_ctx();
    c->s = defs->s;
    c->device = device;
    c->num_cus = prop.multiProcessorCount;
    c->debug = debug_flags_from_env();
    // host-buffer batches (hrx_host_api.cpp): the defaults of HRX_OPT_HOST_PIPELINE and the pipeline's chunk size / trace come from the environment ONCE, here
    if (const char *v = std::getenv("HRX_HOST_PIPELINE")) c->host_pipeline = std::atoi(v) != 0 ? 1 : 2;
    if (const char *v = std::getenv("HRX_HOST_CHUNK_MIB")) { const long n = std::atol(v); if (n >= 4 && n <= 4096) c->host_chunk_mib = (size_t)n; }
    if (const char *v = std::getenv("HRX_HOST_TRACE")) c->host_trace = std::atoi(v) != 0;
    // tuning knobs of the placement search: HRX_PLACE=0 switches it off, HRX_PLACE_MAX_STEPS bounds the walk, HRX_PLACE_TRACE=1 prints every measured step to stderr
    if (const char *v = std::getenv("HRX_PLACE")) c->place_enabled = std::atoi(v) != 0;
    if (const char *v = std::getenv("HRX_MP_COMBINE")) c->mp_combine = std::atoi(v) != 0;
    if (const char *v = std::getenv("HRX_PLACE_TRACE")) c->place_trace = std::atoi(v) != 0;
    if (const char *v = std::getenv("HRX_PLACE_MAX_STEPS")) { const int n = std::atoi(v); if (n >= 1 && n <= 256) { c->place_max_steps = n; c->place_max_steps_set = true; } }
    c->pool = pool_acquire(device);
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_group_counter, 64);
    if (e == hipSuccess) e = hipMemset(c->d_group_counter, 0, 64);

    if (c->s.groups.empty()) {
        if (e == hipSuccess) e = upload_images(c->s, c->d_table, c->d_wide, c->d_half, c->d_pairtab);
        if (e == hipSuccess) e = upload_blob(c->s.byte.image, c->d_bytetab);
    } else {   // multi-pass: the kernels only ever see a group's images
        c->groups.resize(c->s.groups.size());
        for (size_t g = 0; e == hipSuccess && g < c->s.groups.size(); ++g) {
            e = upload_images(c->s.groups[g], c->groups[g].d_table, c->groups[g].d_wide, c->groups[g].d_half, c->groups[g].d_pairtab);
            if (e == hipSuccess) e = upload_blob(c->s.groups[g].byte.image, c->groups[g].d_bytetab);
        }
    }
    c->cw_groups.resize(c->s.cw_groups.size());
    for (size_t g = 0; e == hipSuccess && g < c->s.cw_groups.size(); ++g) {
        e = c->cw_groups[g].d_cw.reserve(c->s.cw_groups[g].cw_image.size());
        if (e == hipSuccess) e = hipMemcpy(c->cw_groups[g].d_cw.p, c->s.cw_groups[g].cw_image.data(), c->s.cw_groups[g].cw_image.size(), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && !c->s.cw_image.empty()) {
        e = c->d_cw.reserve(c->s.cw_image.size());
        if (e == hipSuccess) e = hipMemcpy(c->d_cw.p, c->s.cw_image.data(), c->s.cw_image.size(), hipMemcpyHostToDevice);
    }
    for (size_t d = 0; e == hipSuccess && d < c->s.pair_tags.size(); ++d) {
        uint16_t *p = nullptr;
        e = hipMalloc((void **)&p, c->s.pair_tags[d].size() * 2);
        if (e == hipSuccess) {
            c->d_pair.push_back(p);
            e = hipMemcpy(p, c->s.pair_tags[d].data(), c->s.pair_tags[d].size() * 2, hipMemcpyHostToDevice);
        }
    }
    for (size_t d = 0; e == hipSuccess && d < c->s.endpoint_member.size(); ++d) {
        uint8_t *p = nullptr;
        e = hipMalloc((void **)&p, c->s.endpoint_member[d].size());
        if (e == hipSuccess) {
            c->d_member.push_back(p);
            e = hipMemcpy(p, c->s.endpoint_member[d].data(), c->s.endpoint_member[d].size(), hipMemcpyHostToDevice);
        }
    }
    if (e != hipSuccess) {
        hrx_ctx_destroy(c);
        return fail(HRX_ERR_HIP, std::string("ctx setup: ") + hipGetErrorString(e));
    }
    *out = c;
    return HRX_OK;
}

int hrx_ctx_set_host_threshold(hrx_ctx *ctx, size_t rows) {
    if (!ctx) return fail(HRX_ERR_ARG, "NULL ctx");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->host_threshold = rows;
    return HRX_OK;
}

size_t hrx_ctx_host_threshold(const hrx_ctx *ctx) { return ctx ? ctx->host_threshold : 0; }
int hrx_ctx_device(const hrx_ctx *ctx) { return ctx ? ctx->device : HRX_DEVICE_NONE; }

void hrx_ctx_destroy(hrx_ctx *c) {
    if (!c) return;
    DeviceGuard guard;
    if (c->device != HRX_DEVICE_NONE) (void)guard.set(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    if (c->d_table) (void)hipFree(c->d_table);
    if (c->d_wide) (void)hipFree(c->d_wide);
    if (c->d_half) (void)hipFree(c->d_half);
    if (c->d_pairtab) (void)hipFree(c->d_pairtab);
    if (c->d_bytetab) (void)hipFree(c->d_bytetab);
    for (auto &g : c->groups) {
        if (g.d_table) (void)hipFree(g.d_table);
        if (g.d_wide) (void)hipFree(g.d_wide);
        if (g.d_half) (void)hipFree(g.d_half);
        if (g.d_pairtab) (void)hipFree(g.d_pairtab);
        if (g.d_bytetab) (void)hipFree(g.d_bytetab);
        g.records.release(); g.status.release(); g.summary.release();
    }
    c->mp_masked.release(); c->mp_ov.release();
    c->tp_records.release(); c->tp_masked.release();
    c->spec_cimage.release(); c->d_cw.release();
    for (auto &g : c->cw_groups) { g.d_cw.release(); g.status.release(); g.summary.release(); }
    c->spec_cls.release(); c->spec_ends.release(); c->spec_fail.release(); c->spec_init.release(); c->spec_vstatus.release(); c->spec_vinfo.release(); c->spec_work.release();
    if (c->d_group_counter) (void)hipFree(c->d_group_counter);
    pool_release(c->pool);   // the device's arena pair goes with its last context (now, or with its last sub-buffer)
#ifdef HRX_STAMPS
    c->stamps.release();
#endif
    for (uint16_t *p : c->d_pair) (void)hipFree(p);
    for (uint8_t *p : c->d_member) (void)hipFree(p);
    c->chars.release(); c->lens.release(); c->records.release(); c->masked.release();
    c->status.release(); c->states.release(); c->tags.release();
    delete c;
}

/* ------------------------------ the hot path ------------------------------ */

}  // extern "C"

int launch_batch(hrx_ctx *ctx, const uint8_t *chars, size_t stride, const uint32_t *lens, size_t B, size_t M,
                 uint32_t *records, uint16_t *masked, uint64_t *status, hipStream_t st, size_t rec_pitch,
                 size_t msk_pitch, int layout, uint32_t *const *planes, size_t n_planes) {
    // planes: the D record planes in buffers of their own (hrx_witness_batch_device_planes; position-major outputs, a config that runs as ONE launch); records = planes[0] then
    if (!rec_pitch) rec_pitch = M;
    if (!msk_pitch) msk_pitch = M;
    if (ctx->device == HRX_DEVICE_NONE) return fail(HRX_ERR_HIP, "host-only context (HRX_DEVICE_NONE): no device to launch on");
    if (B == 0) return HRX_OK;
    if (!chars || !lens || !records || !masked || !status) return fail(HRX_ERR_ARG, "NULL buffer");
    if (M == 0 || M > (1u << 24)) return fail(HRX_ERR_ARG, "max_chars_size must be in 1..2^24");
    if (B > 0xffffffffull - 64) return fail(HRX_ERR_ARG, "batch too large");
    if ((stride & 15) || stride < 16 || ((uintptr_t)chars & 15))
        return fail(HRX_ERR_ARG, "chars must be 16-byte aligned with stride % 16 == 0 and stride >= 16");
    if (((uintptr_t)records & 15) || ((uintptr_t)masked & 15) || ((uintptr_t)status & 7) || ((uintptr_t)lens & 3))
        return fail(HRX_ERR_ARG, "output buffers must be 16-byte aligned");
    if (rec_pitch < M || msk_pitch < M || rec_pitch > (1u << 24) || msk_pitch > (1u << 24))
        return fail(HRX_ERR_ARG, "row pitches must be >= max_chars_size");
    if ((M % 8 == 0) && ((rec_pitch % 8) || (msk_pitch % 8)))
        return fail(HRX_ERR_ARG, "row pitches must be multiples of 8 rows when max_chars_size is");
    if (layout != HRX_LAYOUT_STRING_MAJOR && layout != HRX_LAYOUT_POSITION_MAJOR &&
        layout != (HRX_LAYOUT_POSITION_MAJOR | HRX_LAYOUT_INPUT_POSITION_MAJOR))
        return fail(HRX_ERR_ARG, "unknown layout");
    // one launch over `set` (a config of up to kMaxDefsPerPass defs, or one group of a larger one) with that set's device images
    const uint16_t *const *d_pair_tags = ctx->s.groups.empty() && !ctx->d_pair.empty() ? ctx->d_pair.data() : nullptr;
    auto launch_set = [&](const DefsSet &set, const uint32_t *d_table, const uint64_t *d_wide, const uint16_t *d_half, const uint8_t *d_pairtab,
                          const uint8_t *d_bytetab, int lay, uint32_t *rec, uint16_t *msk, uint64_t *stat, size_t rp, size_t mp,
                          uint32_t rec_D = 0, uint32_t rec_d0 = 0, uint32_t *summary = nullptr,
                          uint32_t merge_G = 0, const uint32_t *const *merge_summary = nullptr, uint32_t *merge_ov = nullptr) -> int {
        WitnessArgs a{};
        a.rec_D = rec_D; a.rec_d0 = rec_d0; a.summary = summary;
        a.merge_G = merge_G; a.merge_ov = merge_ov;
        for (uint32_t g = 0; g < merge_G && g < kMaxMergeGroups; ++g) a.merge_summary[g] = merge_summary[g];
        const bool pass = summary != nullptr || merge_G != 0;    // a pass of a multi-pass config
        a.rec_pitch = (uint32_t)rp; a.msk_pitch = (uint32_t)mp;
        a.layout = (uint32_t)lay;
        a.chars = chars; a.stride = stride; a.lens = lens; a.B = (uint32_t)B; a.M = (uint32_t)M;
        a.records = rec; a.masked = msk; a.status = stat;
        a.table_image = d_table; a.table_bytes = (uint32_t)(set.table_image.size() * 4);
        a.wide_image = d_wide;
        a.half_image = d_half; a.half_bytes = (uint32_t)(set.half_image.size() * 2);
        a.pair_image = d_pairtab; a.pair_bytes = set.pair.bytes; a.pair_classes = set.pair.n_classes;
        a.pair_blk_bytes = set.pair.blk_bytes; a.pair_lut_off = set.pair.lut_off;
        a.byte_image = d_bytetab; a.byte_bytes = set.byte.bytes; a.byte_ptab_off = set.byte.ptab_off; a.byte_mul_a4 = set.byte.mul_a * 4; a.byte_mul_b4 = set.byte.mul_b * 4; a.byte_slot_mask4 = (set.byte.slots - 1u) * 4u;
        a.byte_dead = set.byte.dead;
        a.byte_rows_bytes = set.byte.n_rows * 256u; a.byte16_bytes = set.byte.bytes16; a.byte16_ptab_off = set.byte.ptab16_off;
        a.byte_one_id = (set.defs.size() == 1 && set.defs[0].substrs.size() == 1 && set.consts[0].substr_id_offset >= 1 && set.consts[0].substr_id_offset <= 63) ? set.consts[0].substr_id_offset : 0u;
        a.D = (uint32_t)set.defs.size();
        a.debug = ctx->debug;
        a.tune = ctx->tune;
#ifdef HRX_ABLATION
        a.debug = debug_flags_from_env();   // tools/ab_flags.py switches ablations between launches of one process
#endif
        if (planes) {
            for (uint32_t d = 0; d < n_planes && d < kMaxDefsPerLaunch; ++d) a.rec_planes[d] = (unsigned char *)planes[d];
            if (n_planes == 2 * (size_t)a.D) {      // one def in two row stripes: the loader / walker / finisher kernel writes them (not the pair-step kernel, and not in chunks:
                a.rec_stripes = 2;                  // a chunk's first tile is not a stripe boundary in general, and the chunked launch's repair reads whole planes)
                a.debug |= kDbgNoPair | kDbgNoSpec;
            }
        }
        // a summary-writing pass is the loader / walker / finisher kernel: no pair-step or def-parallel variant, no HALF table
        if (pass) a.debug |= kDbgNoPair | kDbgNoDefParallel;
        for (uint32_t d = 0; d < a.D && d < kMaxDefsPerLaunch; ++d) a.dc[d] = set.consts[d];
        LaunchInfo li;
        if (!plan_witness_launch(a, ctx->num_cus, li)) return fail(HRX_ERR_BOUNDS, "tables + staging do not fit the 160 KiB LDS");
        if (pass && li.half) {   // (a group of one big DFA: walk its 4-byte table out of L2 instead)
            a.debug |= kDbgForceGlobalTable;
            if (!plan_witness_launch(a, ctx->num_cus, li)) return fail(HRX_ERR_BOUNDS, "tables + staging do not fit the 160 KiB LDS");
        }
        if (pass && li.split != 2) return fail(HRX_ERR_STATE, "multi-pass: the planner did not pick the position-major loader/walker kernel");
        SpecArgs sp{};
        if (li.spec_tiles) {
            // the chunked launch's state (quasi-absorbing states, pair-tag tables) belongs to the WHOLE config: a group of a multi-pass config is never chunked
            // (today because a summary / merge pass is excluded by the planner; made explicit so that a planner change cannot walk into a NULL table)
            if (&set != &ctx->s || !d_pair_tags) return fail(HRX_ERR_STATE, "chunked launch planned for a group of a multi-pass config");
            // ---- chunked launch (hrx_kernel_spec.hip): scout + compose find every chunk's start state, the walk below runs over the
            // chunks, the stitch launch behind it settles what crosses the chunk borders.  Context scratch, like the group buffers.
            if (ctx->scratch_used && ctx->scratch_stream != st && hipStreamSynchronize(ctx->scratch_stream) != hipSuccess) {
                (void)hipGetLastError();
                return fail(HRX_ERR_STATE, "the context's chunk scratch is in use on another stream and that stream cannot be waited for here (stream capture?): one context serves one stream / graph at a time");
            }
            ctx->scratch_stream = st; ctx->scratch_used = true;
            const uint32_t C = (uint32_t)li.spec_chunks, Dn = a.D, G = a.n_groups;
            const size_t Bpad = (size_t)G * 64;
            uint32_t smax = 4;
            for (uint32_t d = 0; d < Dn; ++d) smax = std::max<uint32_t>(smax, (uint32_t)set.defs[d].allstr.largest_state_val + 1);
            smax = (smax + 15u) & ~15u;
            const uint32_t row_bytes = smax + 32u;
            HIP_TRY(ctx->spec_cls.reserve((size_t)C * Dn * Bpad * row_bytes));   // one row per (chunk, def, string)
            HIP_TRY(ctx->spec_init.reserve((size_t)C * B * Dn * 4));
            HIP_TRY(ctx->spec_vstatus.reserve((size_t)C * B * 8));
            HIP_TRY(ctx->spec_vinfo.reserve((size_t)C * B * 8));
            HIP_TRY(ctx->spec_work.reserve(16 + (size_t)C * B * 8));
            sp.chars = chars; sp.stride = stride; sp.lens = lens; sp.B = (uint32_t)B; sp.M = (uint32_t)M; sp.D = Dn; sp.in_pm = (lay & HRX_LAYOUT_INPUT_POSITION_MAJOR) ? 1u : 0u;
            sp.C = C; sp.tiles_per_chunk = (uint32_t)li.spec_tiles; sp.n_groups = G;
            sp.table_image = d_table; sp.table_bytes = a.table_bytes;
            sp.smax = smax;
            for (uint32_t d = 0; d < Dn; ++d) {
                sp.dc[d] = set.consts[d];
                sp.n_states[d] = (uint32_t)set.defs[d].allstr.largest_state_val + 1;
                sp.pair_tags[d] = d_pair_tags ? d_pair_tags[d] : nullptr;
            }
            if (!ctx->spec_qabs_ready) {     // quasi-absorbing states (hrx_kernel_spec.hip), once per context: from the defs' dense tables
                std::memset(ctx->spec_qabs, 0, sizeof ctx->spec_qabs);
                for (uint32_t d = 0; d < Dn; ++d) {
                    const uint32_t S = sp.n_states[d], *T = set.table_image.data() + (size_t)set.consts[d].row_base * 256;
                    bool undefined_everywhere[256];
                    for (uint32_t c = 0; c < 256; ++c) {
                        undefined_everywhere[c] = true;
                        for (uint32_t t = 0; t < S && undefined_everywhere[c]; ++t) undefined_everywhere[c] = T[(size_t)t * 256 + c] >= set.consts[d].dead_entry;
                    }
                    for (uint32_t t = 0; t < S && t < 256; ++t) {
                        bool q = true;
                        for (uint32_t c = 0; c < 256 && q; ++c) {
                            const uint32_t e = T[(size_t)t * 256 + c];
                            q = e >= set.consts[d].dead_entry ? undefined_everywhere[c] : (e >> kNextShift) - set.consts[d].row_base == t;
                        }
                        if (q) ctx->spec_qabs[d][t >> 5] |= 1u << (t & 31);
                    }
                }
                // ---- the scout's compact tables: per def, bytes with the same column in every row (real states, dummy row, dead row) are one CLASS
                std::vector<uint8_t> img;
                bool compact = true;
                for (uint32_t d = 0; d < Dn && compact; ++d) {
                    const uint32_t S = sp.n_states[d], R = S + 2u, base = set.consts[d].row_base;
                    const uint32_t *T = set.table_image.data() + (size_t)base * 256;
                    auto rel = [&](uint32_t r, uint32_t c) { const uint32_t e = T[(size_t)r * 256 + c]; const uint32_t x = (e >> kNextShift) - base; return e >= set.consts[d].dead_entry || x > S + 1u ? S + 1u : x; };
                    std::map<std::vector<uint16_t>, uint32_t> cls_of;
                    std::vector<std::vector<uint16_t>> cols;
                    uint8_t lut[256];
                    for (uint32_t c = 0; c < 256 && compact; ++c) {
                        std::vector<uint16_t> col(R);
                        for (uint32_t r = 0; r < R; ++r) col[r] = (uint16_t)rel(r, c);
                        auto it = cls_of.find(col);
                        if (it == cls_of.end()) {
                            if (cols.size() >= 127) { compact = false; break; }
                            it = cls_of.emplace(col, (uint32_t)cols.size()).first;
                            cols.push_back(col);
                        }
                        lut[c] = (uint8_t)(it->second * 2u);
                    }
                    if (!compact) break;
                    const uint32_t Cn = (uint32_t)cols.size(), rowb = Cn * 2u;
                    const uint32_t lut_off = (uint32_t)img.size();
                    img.insert(img.end(), lut, lut + 256);
                    const uint32_t tab_off = (uint32_t)img.size();
                    if ((size_t)tab_off + (size_t)R * rowb > 0xfff0u) { compact = false; break; }      // row addresses are u16
                    img.resize((size_t)tab_off + (((size_t)R * rowb + 15) & ~(size_t)15), 0);
                    for (uint32_t r = 0; r < R; ++r)
                        for (uint32_t k = 0; k < Cn; ++k) {
                            const uint16_t addr = (uint16_t)(tab_off + (uint32_t)cols[k][r] * rowb);
                            std::memcpy(&img[(size_t)tab_off + (size_t)r * rowb + 2u * k], &addr, 2);
                        }
                    ctx->spec_c_lut[d] = lut_off; ctx->spec_c_tab[d] = tab_off; ctx->spec_c_rowb[d] = rowb; ctx->spec_c_inv[d] = (65536u + rowb - 1u) / rowb;
                }
                ctx->spec_cimage_bytes = 0;
                if (compact && !img.empty() && !std::getenv("HRX_SPEC_NO_COMPACT")) {
                    HIP_TRY(ctx->spec_cimage.reserve(img.size()));
                    HIP_TRY(hipMemcpy(ctx->spec_cimage.p, img.data(), img.size(), hipMemcpyHostToDevice));       // (once per context, like the allocations above)
                    ctx->spec_cimage_bytes = (uint32_t)img.size();
                }
                ctx->spec_qabs_ready = true;
            }
            std::memcpy(sp.qabs, ctx->spec_qabs, sizeof sp.qabs);
            sp.cimage = ctx->spec_cimage_bytes ? (const uint8_t *)ctx->spec_cimage.p : nullptr; sp.cimage_bytes = ctx->spec_cimage_bytes;
            for (uint32_t d = 0; d < Dn && d < kMaxDefsPerPass; ++d) { sp.c_lut[d] = ctx->spec_c_lut[d]; sp.c_tab[d] = ctx->spec_c_tab[d]; sp.c_rowb[d] = ctx->spec_c_rowb[d]; sp.c_inv[d] = ctx->spec_c_inv[d]; }
            sp.rows = (uint8_t *)ctx->spec_cls.p; sp.row_bytes = row_bytes;
#ifdef HRX_ABLATION
            if (const char *v = std::getenv("HRX_SPEC_DBG")) sp.dbg = (uint32_t)std::strtoul(v, nullptr, 0);
#endif
            sp.init = (uint32_t *)ctx->spec_init.p; sp.vinfo = (const uint2 *)ctx->spec_vinfo.p; sp.vstatus = (const uint64_t *)ctx->spec_vstatus.p;
            sp.status = stat; sp.records = rec; sp.masked = msk;
            if (planes)
                for (uint32_t d = 0; d < Dn && d < kMaxDefsPerPass; ++d) sp.rec_planes[d] = planes[d];
            sp.work_count = (uint32_t *)ctx->spec_work.p; sp.work = (uint2 *)((unsigned char *)ctx->spec_work.p + 16); sp.work_cap = (uint32_t)(C * B);
            HIP_TRY(launch_spec_scout(sp, ctx->num_cus, st));
            HIP_TRY(launch_spec_compose(sp, st));
            a.vs_init = sp.init; a.vs_chunks = C; a.vs_tiles = (uint32_t)li.spec_tiles; a.vs_groups = G;
            a.vs_status = (uint64_t *)ctx->spec_vstatus.p; a.vs_info = (uint2 *)ctx->spec_vinfo.p;
            a.n_groups = G * C;
        }
        if (li.dyn) {   // launches that share the counter must not overlap: a launch on another stream waits for the previous one
            if (ctx->scratch_used && ctx->scratch_stream != st && hipStreamSynchronize(ctx->scratch_stream) != hipSuccess) {
                (void)hipGetLastError();
                return fail(HRX_ERR_STATE, "the context's launch scratch is in use on another stream and that stream cannot be waited for here (stream capture?): one context serves one stream / graph at a time");
            }
            ctx->scratch_stream = st; ctx->scratch_used = true;
            const uint32_t slots = (uint32_t)li.grid * (uint32_t)(li.waves_per_wg / ((a.layout & 1u) && !li.half ? 3 : 2));
            // (a kernel, not hipMemsetAsync: captured into a HIP graph, memset nodes were seen executing out of order with the kernel nodes
            // around them — replays whose first and last launch found the counter exhausted and skipped their dynamic groups,
            // tools/dyn_graph_debug.py)
            HIP_TRY(launch_zero_u32(ctx->d_group_counter, st));
            a.group_counter = ctx->d_group_counter;
            a.group_base = 0;
            a.group_first_dyn = slots;
        }
        a.nt_mix = plan_nt_mix(a, li);
#if defined(HRX_STAMPS) || defined(HRX_ABLATION) || defined(HRX_NT_ENV)
        if (const char *nm = std::getenv("HRX_NT_MIX")) a.nt_mix = (uint32_t)std::strtoul(nm, nullptr, 0);
        if (const char *nf = std::getenv("HRX_NT_FLAGS")) a.nt_mix |= (uint32_t)std::strtoul(nf, nullptr, 0);   // (kNtMix* bits, hrx_kernel.hpp)
#endif
#if defined(HRX_STAMPS) || defined(HRX_ABLATION)
        if (const char *pe = std::getenv("HRX_PACE")) a.pace_even = (uint32_t)std::strtoul(pe, nullptr, 0);
#endif
#ifdef HRX_STAMPS
        if (li.split == 2 || li.split == 5) {
            const size_t bytes = (size_t)li.grid * 8 * 16 * 8;   // 16 u64 per walker pair (8 walker + 8 finisher stamps), up to 8 pairs per workgroup
            if (ctx->stamps.cap < bytes) { HIP_TRY(ctx->stamps.reserve(bytes)); }
            HIP_TRY(hipMemsetAsync(ctx->stamps.p, 0, bytes, st));
            a.stamps = (unsigned long long *)ctx->stamps.p;
        }
#endif
        HIP_TRY(launch_witness(a, li, st));
        if (li.spec_tiles) {
            HIP_TRY(launch_spec_stitch(sp, ctx->num_cus, st));
        }
        return HRX_OK;
    };
    if (planes && !ctx->s.groups.empty() && !(ctx->d_cw.p && !ctx->mp_combine))
        return fail(HRX_ERR_ARG, "record planes: configs that run as one launch (up to three defs, or four to eight defs of at most 32 byte classes each)");
    if (ctx->s.groups.empty()) {
        // string-major outputs of a DFA whose 4-byte table does not fit LDS (cfg 5): the string-major kernels would walk it out of global
        // memory (0.21 of peak); the BYTE / HALF table kernels are position-major — run them into context scratch and turn the rows around
        // (one def with a BYTE image has a string-major kernel of its own: the walker/storer kernel on that table)
        const bool byte_split = !ctx->s.byte.image.empty() && !(ctx->debug & (kDbgNoByte | kDbgForceHalf));
        if (layout == HRX_LAYOUT_STRING_MAJOR && M % 8 == 0 && !byte_split &&
            (!ctx->s.byte.image.empty() || !ctx->s.half_image.empty()) && ctx->s.table_image.size() * 4 + wave_stage_bytes((int)ctx->s.defs.size(), 16) > kLdsLimit) {
            if (ctx->scratch_used && ctx->scratch_stream != st && hipStreamSynchronize(ctx->scratch_stream) != hipSuccess) {
                (void)hipGetLastError();
                return fail(HRX_ERR_STATE, "the context's transpose scratch is in use on another stream and that stream cannot be waited for here (stream capture?): one context serves one stream / graph at a time");
            }
            ctx->scratch_stream = st; ctx->scratch_used = true;
            const size_t q4 = (M + 3) / 4, q8 = (M + 7) / 8, D = ctx->s.defs.size();
            HIP_TRY(ctx->tp_records.reserve(q4 * 4 * D * B * 4));
            HIP_TRY(ctx->tp_masked.reserve(q8 * 8 * B * 2));
            const int rc = launch_set(ctx->s, ctx->d_table, ctx->d_wide, ctx->d_half, ctx->d_pairtab, ctx->d_bytetab, HRX_LAYOUT_POSITION_MAJOR,
                                      (uint32_t *)ctx->tp_records.p, (uint16_t *)ctx->tp_masked.p, status, M, M);
            if (rc != HRX_OK) return rc;
            TransposeArgs ta{(const uint32_t *)ctx->tp_records.p, (const uint16_t *)ctx->tp_masked.p, (uint32_t)B, (uint32_t)M, (uint32_t)D, records, masked, (uint32_t)rec_pitch, (uint32_t)msk_pitch};
            HIP_TRY(launch_transpose(ta, st));
            return HRX_OK;
        }
        return launch_set(ctx->s, ctx->d_table, ctx->d_wide, ctx->d_half, ctx->d_pairtab, ctx->d_bytetab, layout, records, masked, status, rec_pitch, msk_pitch);
    }
    // ---- more than kMaxDefsPerPass defs: one ordinary launch per group into its private position-major buffers, then the
    // combine kernel (hrx_kernel_mp.hip) writes the caller's buffers.  The group buffers belong to the context: a launch
    // on another stream first waits for the previous combine.
    const size_t G = ctx->s.groups.size();
    if (G > kMaxGroups) return fail(HRX_ERR_BOUNDS, "too many def groups");
    if (ctx->scratch_used && ctx->scratch_stream != st && hipStreamSynchronize(ctx->scratch_stream) != hipSuccess) {
        (void)hipGetLastError();
        return fail(HRX_ERR_STATE, "the context's group buffers are in use on another stream and that stream cannot be waited for here (stream capture?): one context serves one stream / graph at a time");
    }
    ctx->scratch_stream = st; ctx->scratch_used = true;
    const size_t q4 = (M + 3) / 4, q8 = (M + 7) / 8, ntiles = (M + 63) / 64;
    // string-major outputs (rows in multiples of 8): the passes and the combine run position-major into context scratch and
    // transpose_pm_to_sm_kernel turns the rows around (hrx_kernel_tp.hip: 3.7 -> ~1 ms at D = 5, 65536 x 1024 rows); other row counts keep the
    // copy-mode combine (per-lane 4-byte stores)
    // ---- four and five defs, string-major outputs, rows in multiples of 16: the def-parallel launch writes the caller's [B][pitch][D] records and [B][pitch] masked rows itself — its walkers'
    // quads meet in LDS sub-tiles, a storer wave writes each string's 16 rows x D records as one run (hrx_kernel_pmd.hip SMO) — instead of position-major scratch + the transposer
    if (layout == HRX_LAYOUT_STRING_MAJOR && ctx->d_cw.p && !ctx->mp_combine) {
        WitnessArgs a{};
        a.layout = (uint32_t)layout; a.chars = chars; a.stride = stride; a.lens = lens; a.B = (uint32_t)B; a.M = (uint32_t)M;
        a.rec_pitch = (uint32_t)rec_pitch; a.msk_pitch = (uint32_t)msk_pitch;
        a.records = records; a.masked = masked; a.status = status;
        a.D = (uint32_t)ctx->s.defs.size();
        a.debug = ctx->debug;
        a.cw_image = (const uint8_t *)ctx->d_cw.p; a.cw_lut_off = ctx->s.cw_lut_off; a.table_bytes = (uint32_t)ctx->s.cw_image.size();
        for (uint32_t d = 0; d < a.D && d < kMaxDefsPerLaunch; ++d) a.dc[d] = ctx->s.cw_consts[d];
        LaunchInfo li{};
        if (plan_pmd_cw_sm(a, ctx->num_cus, li)) {
            a.nt_mix = plan_nt_mix(a, li);
            HIP_TRY(launch_witness(a, li, st));
            return HRX_OK;
        }
    }
    const bool via_tp = !(layout & HRX_LAYOUT_POSITION_MAJOR) && M % 8 == 0;
    uint32_t *const caller_records = records;
    uint16_t *const caller_masked = masked;
    if (via_tp) {
        HIP_TRY(ctx->tp_records.reserve(q4 * 4 * ctx->s.defs.size() * B * 4));
        HIP_TRY(ctx->tp_masked.reserve(q8 * 8 * B * 2));
        records = (uint32_t *)ctx->tp_records.p; masked = (uint16_t *)ctx->tp_masked.p;
        layout = HRX_LAYOUT_POSITION_MAJOR | (layout & HRX_LAYOUT_INPUT_POSITION_MAJOR);
    }
    // ---- 6 or 7 defs whose CLASS-WIDE tables exist (every def <= 32 byte classes; 4 and 5 when forced): ONE def-parallel launch walks all defs — one walker wave per def, the last one combining
    // (hrx_kernel_pmd.hip CW) — instead of passes over groups of three with summaries in between: no second read of the input, no 80-byte tile summaries written and read back
    if ((layout & HRX_LAYOUT_POSITION_MAJOR) && ctx->d_cw.p && !ctx->mp_combine) {
        WitnessArgs a{};
        a.layout = (uint32_t)layout; a.chars = chars; a.stride = stride; a.lens = lens; a.B = (uint32_t)B; a.M = (uint32_t)M;
        a.rec_pitch = (uint32_t)M; a.msk_pitch = (uint32_t)M;
        a.records = records; a.masked = masked; a.status = status;
        a.D = (uint32_t)ctx->s.defs.size();
        a.debug = ctx->debug;
        a.cw_image = (const uint8_t *)ctx->d_cw.p; a.cw_lut_off = ctx->s.cw_lut_off; a.table_bytes = (uint32_t)ctx->s.cw_image.size();
        for (uint32_t d = 0; d < a.D && d < kMaxDefsPerLaunch; ++d) a.dc[d] = ctx->s.cw_consts[d];
        if (planes)
            for (uint32_t d = 0; d < a.D && d < kMaxDefsPerLaunch; ++d) a.rec_planes[d] = (unsigned char *)planes[d];
        LaunchInfo li{};
        if (plan_pmd_cw(a, ctx->num_cus, li)) {
            a.nt_mix = plan_nt_mix(a, li);
            HIP_TRY(launch_witness(a, li, st));
            if (via_tp) {
                TransposeArgs ta{records, masked, (uint32_t)B, (uint32_t)M, (uint32_t)ctx->s.defs.size(), caller_records, caller_masked, (uint32_t)rec_pitch, (uint32_t)msk_pitch};
                HIP_TRY(launch_transpose(ta, st));
            }
            return HRX_OK;
        }
    }
    if (planes) return fail(HRX_ERR_BOUNDS, "record planes: the one-launch def-parallel path does not fit this config");
    // ---- more than eight defs, every def of at most 32 byte classes: passes over CW GROUPS of 4 .. 8 defs (DefsSet::cw_groups), each ONE def-parallel launch that writes its defs' planes of the
    // caller's records and a tile summary; the combine launch forms what needs all defs of a row.  D = 16: two passes instead of six, D = 32: four instead of eleven.
    if ((layout & HRX_LAYOUT_POSITION_MAJOR) && !ctx->cw_groups.empty() && !(ctx->debug & kDbgNoDefParallel)) {
        const size_t GC = ctx->cw_groups.size();
        CombineArgs cc{};
        bool planned = true;
        std::vector<LaunchInfo> lis(GC);
        std::vector<WitnessArgs> was(GC);
        for (size_t g = 0; g < GC && planned; ++g) {
            const DefsSet &gs = ctx->s.cw_groups[g];
            WitnessArgs &a = was[g];
            a = WitnessArgs{};
            a.layout = (uint32_t)layout; a.chars = chars; a.stride = stride; a.lens = lens; a.B = (uint32_t)B; a.M = (uint32_t)M;
            a.rec_pitch = (uint32_t)M; a.msk_pitch = (uint32_t)M;
            a.records = records; a.masked = masked;
            a.D = (uint32_t)gs.defs.size();
            a.rec_D = (uint32_t)ctx->s.defs.size(); a.rec_d0 = ctx->s.cw_group_first[g];
            a.debug = ctx->debug;
            a.cw_image = (const uint8_t *)ctx->cw_groups[g].d_cw.p; a.cw_lut_off = gs.cw_lut_off; a.table_bytes = (uint32_t)gs.cw_image.size();
            for (uint32_t d = 0; d < a.D && d < kMaxDefsPerLaunch; ++d) a.dc[d] = gs.cw_consts[d];
            planned = plan_pmd_cw(a, ctx->num_cus, lis[g]);
        }
        if (planned) {
            for (size_t g = 0; g < GC; ++g) {
                hrx_ctx::CwGroupDev &gd = ctx->cw_groups[g];
                HIP_TRY(gd.status.reserve(B * 8));
                HIP_TRY(gd.summary.reserve(ntiles * 5 * B * 16));
                was[g].status = (uint64_t *)gd.status.p;
                was[g].summary = (uint32_t *)gd.summary.p;
                was[g].nt_mix = plan_nt_mix(was[g], lis[g]);
                HIP_TRY(launch_witness(was[g], lis[g], st));
                cc.gsummary[g] = (const uint32_t *)gd.summary.p;
                cc.gstatus[g] = (const uint64_t *)gd.status.p;
                cc.gD[g] = (uint8_t)ctx->s.cw_groups[g].defs.size();
                cc.gfirst[g] = (uint8_t)ctx->s.cw_group_first[g];
            }
            cc.chars = chars; cc.stride = stride; cc.lens = lens; cc.B = (uint32_t)B; cc.M = (uint32_t)M;
            cc.D = (uint32_t)ctx->s.defs.size(); cc.G = (uint32_t)GC; cc.layout = (uint32_t)layout;
            cc.rec_pitch = (uint32_t)rec_pitch; cc.msk_pitch = (uint32_t)msk_pitch;
            cc.records = records; cc.masked = masked; cc.status = status;
            HIP_TRY(launch_combine(cc, st));
            if (via_tp) {
                TransposeArgs ta{records, masked, (uint32_t)B, (uint32_t)M, (uint32_t)ctx->s.defs.size(), caller_records, caller_masked, (uint32_t)rec_pitch, (uint32_t)msk_pitch};
                HIP_TRY(launch_transpose(ta, st));
            }
            return HRX_OK;
        }
    }
    const bool summary_mode = (layout & HRX_LAYOUT_POSITION_MAJOR) != 0;   // position-major outputs: the passes write the caller's record planes themselves
    if (!summary_mode) HIP_TRY(ctx->mp_masked.reserve(q8 * 8 * B * 2));
    // position-major outputs, up to kMaxMergeGroups + 1 groups: the LAST pass reads the earlier groups' summaries itself and writes the final
    // masked rows; a one-thread-per-string launch merges the status words.  (HRX_MP_COMBINE=1: the separate combine launch, as with more groups.)
    const bool merge_last = summary_mode && G - 1 <= kMaxMergeGroups && !ctx->mp_combine;
    if (merge_last) HIP_TRY(ctx->mp_ov.reserve(B * 4));
    CombineArgs ca{};
    for (size_t g = 0; g < G; ++g) {
        const DefsSet &gs = ctx->s.groups[g];
        hrx_ctx::GroupDev &gd = ctx->groups[g];
        HIP_TRY(gd.status.reserve(B * 8));
        int rc;
        if (summary_mode && merge_last && g + 1 == G) {
            const uint32_t *ms[kMaxMergeGroups] = {nullptr, nullptr, nullptr};
            for (size_t e = 0; e + 1 < G; ++e) ms[e] = ca.gsummary[e];
            rc = launch_set(gs, gd.d_table, gd.d_wide, gd.d_half, gd.d_pairtab, gd.d_bytetab, layout, records, masked, (uint64_t *)gd.status.p, M, M,
                            (uint32_t)ctx->s.defs.size(), ctx->s.group_first[g], nullptr, (uint32_t)(G - 1), ms, (uint32_t *)ctx->mp_ov.p);
        } else if (summary_mode) {
            HIP_TRY(gd.summary.reserve(ntiles * 5 * B * 16));
            rc = launch_set(gs, gd.d_table, gd.d_wide, gd.d_half, gd.d_pairtab, gd.d_bytetab, layout, records, masked, (uint64_t *)gd.status.p, M, M,
                            (uint32_t)ctx->s.defs.size(), ctx->s.group_first[g], (uint32_t *)gd.summary.p);
            ca.gsummary[g] = (const uint32_t *)gd.summary.p;
        } else {
            HIP_TRY(gd.records.reserve(q4 * 4 * gs.defs.size() * B * 4));
            rc = launch_set(gs, gd.d_table, gd.d_wide, gd.d_half, gd.d_pairtab, gd.d_bytetab, HRX_LAYOUT_POSITION_MAJOR | (layout & HRX_LAYOUT_INPUT_POSITION_MAJOR),
                            (uint32_t *)gd.records.p, (uint16_t *)ctx->mp_masked.p, (uint64_t *)gd.status.p, M, M);
            ca.grec[g] = (const uint32_t *)gd.records.p;
        }
        if (rc != HRX_OK) return rc;
        ca.gstatus[g] = (const uint64_t *)gd.status.p;
        ca.gD[g] = (uint8_t)gs.defs.size();
        ca.gfirst[g] = (uint8_t)ctx->s.group_first[g];
    }
    ca.chars = chars; ca.stride = stride; ca.lens = lens; ca.B = (uint32_t)B; ca.M = (uint32_t)M;
    ca.D = (uint32_t)ctx->s.defs.size(); ca.G = (uint32_t)G; ca.layout = (uint32_t)layout;
    ca.rec_pitch = (uint32_t)rec_pitch; ca.msk_pitch = (uint32_t)msk_pitch;
    ca.records = records; ca.masked = masked; ca.status = status;
    if (merge_last) {
        ca.merge_ov = (const uint32_t *)ctx->mp_ov.p;
        HIP_TRY(launch_merge_status(ca, st));
    } else {
        HIP_TRY(launch_combine(ca, st));
    }
    if (via_tp) {
        TransposeArgs ta{records, masked, (uint32_t)B, (uint32_t)M, (uint32_t)ctx->s.defs.size(), caller_records, caller_masked, (uint32_t)rec_pitch, (uint32_t)msk_pitch};
        HIP_TRY(launch_transpose(ta, st));
    }
    return HRX_OK;
}

extern "C" {

#ifdef HRX_STAMPS
// tools only (libhrx_stamps.so): the stamps of the LAST position-major launch of this context, 8 u64 per walker pair
int hrx_debug_read_stamps(hrx_ctx *ctx, unsigned long long *out, size_t n_u64) {
    if (!ctx || !out) return HRX_ERR_ARG;
    DeviceGuard guard;
    HIP_TRY(guard.set(ctx->device));
    HIP_TRY(hipDeviceSynchronize());
    const size_t n = n_u64 * 8 < ctx->stamps.cap ? n_u64 * 8 : ctx->stamps.cap;
    HIP_TRY(hipMemcpy(out, ctx->stamps.p, n, hipMemcpyDeviceToHost));
    return HRX_OK;
}
#endif

int hrx_witness_batch_device(hrx_ctx *ctx, const uint8_t *chars, size_t stride, const uint32_t *lens, size_t B, size_t M,
                             uint32_t *records, uint16_t *masked, uint64_t *status, void *stream) {
    if (!ctx) return fail(HRX_ERR_ARG, "NULL ctx");
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard guard;
    if (ctx->device != HRX_DEVICE_NONE) HIP_TRY(guard.set(ctx->device));
    return launch_batch(ctx, chars, stride, lens, B, M, records, masked, status, (hipStream_t)stream);
}

int hrx_witness_batch_device_pitched(hrx_ctx *ctx, const uint8_t *chars, size_t stride, const uint32_t *lens, size_t B, size_t M,
                                     uint32_t *records, size_t rec_pitch, uint16_t *masked, size_t msk_pitch, uint64_t *status,
                                     void *stream) {
    if (!ctx) return fail(HRX_ERR_ARG, "NULL ctx");
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard guard;
    if (ctx->device != HRX_DEVICE_NONE) HIP_TRY(guard.set(ctx->device));
    return launch_batch(ctx, chars, stride, lens, B, M, records, masked, status, (hipStream_t)stream, rec_pitch, msk_pitch);
}

int hrx_witness_batch_device_layout(hrx_ctx *ctx, int layout, const uint8_t *chars, size_t stride, const uint32_t *lens, size_t B,
                                    size_t M, uint32_t *records, uint16_t *masked, uint64_t *status, void *stream) {
    if (!ctx) return fail(HRX_ERR_ARG, "NULL ctx");
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard guard;
    if (ctx->device != HRX_DEVICE_NONE) HIP_TRY(guard.set(ctx->device));
    return launch_batch(ctx, chars, stride, lens, B, M, records, masked, status, (hipStream_t)stream, 0, 0, layout);
}

int hrx_witness_batch_device_planes(hrx_ctx *ctx, int layout, const uint8_t *chars, size_t stride, const uint32_t *lens, size_t B, size_t M,
                                    uint32_t *const *record_planes, size_t n_planes, uint16_t *masked, uint64_t *status, void *stream) {
    if (!ctx) return fail(HRX_ERR_ARG, "NULL ctx");
    if (!(layout & HRX_LAYOUT_POSITION_MAJOR)) return fail(HRX_ERR_ARG, "record planes are a position-major layout");
    const size_t Dn = ctx->s.defs.size();
    if (!record_planes || !(n_planes == Dn || (Dn == 1 && n_planes == 2))) return fail(HRX_ERR_ARG, "record planes: one buffer per RegexDefs of the config (one def: one buffer, or two row stripes)");
    for (size_t d = 0; d < n_planes; ++d)
        if (!record_planes[d] || ((uintptr_t)record_planes[d] & 15)) return fail(HRX_ERR_ARG, "record planes must be 16-byte aligned device buffers");
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard guard;
    if (ctx->device != HRX_DEVICE_NONE) HIP_TRY(guard.set(ctx->device));
    // one def: its plane IS the position-major records buffer
    return launch_batch(ctx, chars, stride, lens, B, M, record_planes[0], masked, status, (hipStream_t)stream, 0, 0, layout, n_planes > 1 ? record_planes : nullptr, n_planes);
}

int hrx_ctx_set_option(hrx_ctx *ctx, int option, long value) {
    if (!ctx) return fail(HRX_ERR_ARG, "NULL ctx");
    std::lock_guard<std::mutex> lk(ctx->mu);
    switch (option) {
        case HRX_OPT_PMD_COMBINER_WAVE:
            if (value < 0 || value > 2) return fail(HRX_ERR_ARG, "HRX_OPT_PMD_COMBINER_WAVE: 0 (default), 1 (on) or 2 (off)");
            ctx->tune = (ctx->tune & ~kTunePmdFinMask) | (uint32_t)value;
            return HRX_OK;
        case HRX_OPT_HOST_ROUTE:
            if (value < 0 || value > 2) return fail(HRX_ERR_ARG, "HRX_OPT_HOST_ROUTE: HRX_HOST_ROUTE_AUTO, _DEVICE or _HOST");
            ctx->host_route = (int)value;
            return HRX_OK;
        case HRX_OPT_HOST_THREADS:
            if (value < 0 || value > 4096) return fail(HRX_ERR_ARG, "HRX_OPT_HOST_THREADS: 0 (the calling thread's cores) .. 4096");
            ctx->host_threads = (int)value;
            return HRX_OK;
        case HRX_OPT_HOST_PIPELINE:
            if (value < 0 || value > 2) return fail(HRX_ERR_ARG, "HRX_OPT_HOST_PIPELINE: 0 (measured), 1 (pipelined) or 2 (one stream)");
            ctx->host_pipeline = (int)value;
            return HRX_OK;
        case HRX_OPT_PLACE_DRY_LAUNCH:
            if (value < 0 || value > 1) return fail(HRX_ERR_ARG, "HRX_OPT_PLACE_DRY_LAUNCH: 0 or 1");
            ctx->place_dry = value != 0;
            return HRX_OK;
        default: return fail(HRX_ERR_ARG, "hrx_ctx_set_option: unknown option");
    }
}

long hrx_ctx_get_option(const hrx_ctx *ctx, int option) {
    if (!ctx) return -1;
    switch (option) {
        case HRX_OPT_PMD_COMBINER_WAVE: return (long)(ctx->tune & kTunePmdFinMask);
        case HRX_OPT_HOST_ROUTE: return ctx->host_route;
        case HRX_OPT_HOST_THREADS: return ctx->host_threads;
        case HRX_OPT_HOST_PIPELINE: return ctx->host_pipeline;
        case HRX_OPT_PLACE_DRY_LAUNCH: return ctx->place_dry ? 1 : 0;
        default: return -1;
    }
}

size_t hrx_fr_num_columns(size_t D) { return 4 + 4 * D; }

void hrx_fr_from_u64(uint64_t v, int flags, uint64_t *limbs) {
    uint64_t w[4];
    if (v >> 32) {
        fr_from_u64(v, w, (flags & HRX_FR_CANONICAL) != 0);
    } else {   // the kernel's route (every witness value fits 32 bits)
        uint32_t h[8];
        fr_from_u32((uint32_t)v, h, (flags & HRX_FR_CANONICAL) != 0);
        for (int i = 0; i < 4; ++i) w[i] = (uint64_t)h[2 * i] | (uint64_t)h[2 * i + 1] << 32;
    }
    if (limbs) { limbs[0] = w[0]; limbs[1] = w[1]; limbs[2] = w[2]; limbs[3] = w[3]; }
}

static int fr_columns_any(hrx_ctx *ctx, int layout, const uint8_t *chars, size_t stride, const uint32_t *lens, const uint32_t *records, const uint32_t *const *planes, size_t n_planes,
                          size_t rec_pitch, const uint16_t *masked, size_t msk_pitch, size_t B, size_t M, size_t b_begin, size_t b_count, uint64_t *cells, int flags, void *stream);

int hrx_fr_columns_device(hrx_ctx *ctx, int layout, const uint8_t *chars, size_t stride, const uint32_t *lens,
                          const uint32_t *records, size_t rec_pitch, const uint16_t *masked, size_t msk_pitch, size_t B,
                          size_t M, size_t b_begin, size_t b_count, uint64_t *cells, int flags, void *stream) {
    return fr_columns_any(ctx, layout, chars, stride, lens, records, nullptr, 0, rec_pitch, masked, msk_pitch, B, M, b_begin, b_count, cells, flags, stream);
}

int hrx_fr_columns_device_planes(hrx_ctx *ctx, int layout, const uint8_t *chars, size_t stride, const uint32_t *lens, const uint32_t *const *record_planes, size_t n_planes,
                                 const uint16_t *masked, size_t B, size_t M, size_t b_begin, size_t b_count, uint64_t *cells, int flags, void *stream) {
    if (!ctx) return fail(HRX_ERR_ARG, "NULL ctx");
    const size_t Dn = ctx->s.defs.size();
    if (!(layout & HRX_LAYOUT_POSITION_MAJOR)) return fail(HRX_ERR_ARG, "record planes are a position-major layout");
    if (!record_planes || !(n_planes == Dn || (Dn == 1 && n_planes == 2)) || n_planes > kMaxDefsPerLaunch) return fail(HRX_ERR_ARG, "record planes: one buffer per def (at most eight; one def: one buffer or two row stripes)");
    for (size_t d = 0; d < n_planes; ++d)
        if (!record_planes[d]) return fail(HRX_ERR_ARG, "NULL plane");
    return fr_columns_any(ctx, layout, chars, stride, lens, record_planes[0], record_planes, n_planes, 0, masked, 0, B, M, b_begin, b_count, cells, flags, stream);
}

static int fr_columns_any(hrx_ctx *ctx, int layout, const uint8_t *chars, size_t stride, const uint32_t *lens, const uint32_t *records, const uint32_t *const *planes, size_t n_planes,
                          size_t rec_pitch, const uint16_t *masked, size_t msk_pitch, size_t B, size_t M, size_t b_begin, size_t b_count, uint64_t *cells, int flags, void *stream) {
    if (!ctx) return fail(HRX_ERR_ARG, "NULL ctx");
    if (b_count == 0) return HRX_OK;
    if (!chars || !lens || !records || !masked || !cells) return fail(HRX_ERR_ARG, "NULL buffer");
    if (b_begin > B || b_count > B - b_begin) return fail(HRX_ERR_ARG, "string range outside the batch");
    if (M == 0 || M > (1u << 24) || B > 0xffffffffull - 64) return fail(HRX_ERR_ARG, "shape out of range");
    if (layout != HRX_LAYOUT_STRING_MAJOR && layout != HRX_LAYOUT_POSITION_MAJOR &&
        layout != (HRX_LAYOUT_POSITION_MAJOR | HRX_LAYOUT_INPUT_POSITION_MAJOR))
        return fail(HRX_ERR_ARG, "unknown layout");
    if ((uintptr_t)cells & 15) return fail(HRX_ERR_ARG, "cells must be 16-byte aligned");
    if (ctx->device == HRX_DEVICE_NONE) return fail(HRX_ERR_HIP, "host-only context (HRX_DEVICE_NONE): no device to launch on");
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard guard;
    HIP_TRY(guard.set(ctx->device));
    FrArgs a{};
    a.chars = chars; a.stride = stride; a.lens = lens; a.records = records; a.masked = masked;
    a.B = (uint32_t)B; a.M = (uint32_t)M; a.D = (uint32_t)ctx->s.defs.size(); a.layout = (uint32_t)layout;
    a.rec_pitch = (uint32_t)(rec_pitch ? rec_pitch : M); a.msk_pitch = (uint32_t)(msk_pitch ? msk_pitch : M);
    a.canonical = (flags & HRX_FR_CANONICAL) ? 1u : 0u;
    if (planes) {
        for (size_t d = 0; d < n_planes && d < kMaxDefsPerLaunch; ++d) a.rec_planes[d] = planes[d];
        a.rec_stripes = n_planes == 2 * ctx->s.defs.size() ? 2u : 1u;
    }
    // one launch per 32768 strings (the grid's y dimension); every launch writes its slice of each column
    const size_t total = b_count;
    for (size_t done = 0; done < total; done += 32768) {
        const size_t nb = total - done < 32768 ? total - done : 32768;
        a.b_begin = (uint32_t)(b_begin + done); a.b_count = (uint32_t)nb;
        a.cells = cells + done * M * 4;
        a.col_cells = total * M;
        HIP_TRY(launch_fr_columns(a, (hipStream_t)stream));
    }
    return HRX_OK;
}
int hrx_chars_to_position_major_device(hrx_ctx *ctx, const uint8_t *chars, size_t stride, size_t B, uint8_t *chars_pm, void *stream) {
    if (!ctx) return fail(HRX_ERR_ARG, "NULL ctx");
    if (ctx->device == HRX_DEVICE_NONE) return fail(HRX_ERR_HIP, "host-only context (HRX_DEVICE_NONE): no device to launch on");
    if (B == 0) return HRX_OK;
    if (!chars || !chars_pm) return fail(HRX_ERR_ARG, "NULL buffer");
    if (B > 0xffffffffull - 64 || stride / 16 > 0xffffffffull) return fail(HRX_ERR_ARG, "shape out of range");
    if ((stride & 15) || stride < 16 || ((uintptr_t)chars & 15) || ((uintptr_t)chars_pm & 15))
        return fail(HRX_ERR_ARG, "buffers must be 16-byte aligned with stride % 16 == 0 and stride >= 16");
    if (chars == chars_pm) return fail(HRX_ERR_ARG, "in-place conversion is not supported");
    DeviceGuard guard;      // (stateless: no context scratch, no lock)
    HIP_TRY(guard.set(ctx->device));
    HIP_TRY(launch_chars_to_position_major(chars, stride, B, chars_pm, (hipStream_t)stream));
    return HRX_OK;
}

void hrx_position_major_plane_sizes(size_t B, size_t M, size_t *plane_u32, size_t *masked_u16) {
    if (plane_u32) *plane_u32 = (M + 3) / 4 * B * 4;
    if (masked_u16) *masked_u16 = (M + 7) / 8 * B * 8;
}

int hrx_rows_of_string_planes(const uint32_t *const *record_planes, size_t n_planes, const uint16_t *masked_pm, size_t B, size_t M, size_t D, size_t b, uint32_t *records, uint16_t *masked) {
    if (!record_planes && !masked_pm) return fail(HRX_ERR_ARG, "NULL buffer");
    if ((record_planes && !records) || (masked_pm && !masked)) return fail(HRX_ERR_ARG, "NULL output");
    if (b >= B || M == 0 || D == 0 || D > HRX_MAX_DEFS) return fail(HRX_ERR_ARG, "string index or shape out of range");
    if (record_planes && !(n_planes == D || (D == 1 && n_planes == 2))) return fail(HRX_ERR_ARG, "one buffer per def (one def: one buffer or two row stripes)");
    const size_t k = b / HRX_PM_BLOCK, bl = b % HRX_PM_BLOCK, nb = std::min<size_t>(HRX_PM_BLOCK, B - k * HRX_PM_BLOCK);
    const size_t q4 = (M + 3) / 4, R = record_planes ? n_planes / D : 1, slots = (q4 + R - 1) / R;
    if (record_planes) {
        for (size_t p = 0; p < n_planes; ++p)
            if (!record_planes[p]) return fail(HRX_ERR_ARG, "NULL plane");
        for (size_t d = 0; d < D; ++d)
            for (size_t q = 0; q < q4; ++q) {      // quad q of def d: buffer (q % R) * D + d, slot q / R
                const uint32_t *src = record_planes[(q % R) * D + d] + k * HRX_PM_BLOCK * slots * 4 + ((q / R) * nb + bl) * 4;
                const size_t rows = std::min<size_t>(4, M - 4 * q);
                for (size_t i = 0; i < rows; ++i) records[(4 * q + i) * D + d] = src[i];
            }
    }
    if (masked_pm) return hrx_rows_of_string_position_major(nullptr, masked_pm, B, M, D, b, nullptr, masked);
    return HRX_OK;
}

int hrx_rows_of_string_position_major(const uint32_t *records_pm, const uint16_t *masked_pm, size_t B, size_t M, size_t D, size_t b, uint32_t *records, uint16_t *masked) {
    if (!records_pm && !masked_pm) return fail(HRX_ERR_ARG, "NULL buffer");
    if ((records_pm && !records) || (masked_pm && !masked)) return fail(HRX_ERR_ARG, "NULL output");
    if (b >= B || M == 0 || D == 0 || D > HRX_MAX_DEFS) return fail(HRX_ERR_ARG, "string index or shape out of range");
    const size_t k = b / HRX_PM_BLOCK, bl = b % HRX_PM_BLOCK, nb = std::min<size_t>(HRX_PM_BLOCK, B - k * HRX_PM_BLOCK);
    const size_t q4 = (M + 3) / 4, q8 = (M + 7) / 8;
    if (records_pm) {
        const uint32_t *base = records_pm + k * HRX_PM_BLOCK * q4 * D * 4 + bl * 4;       // quad q of def d: base + (q * D + d) * nb * 4
        for (size_t q = 0; q < q4; ++q) {
            const size_t rows = std::min<size_t>(4, M - 4 * q);
            for (size_t d = 0; d < D; ++d) {
                const uint32_t *src = base + (q * D + d) * nb * 4;
                for (size_t i = 0; i < rows; ++i) records[(4 * q + i) * D + d] = src[i];
            }
        }
    }
    if (masked_pm) {
        const uint16_t *base = masked_pm + k * HRX_PM_BLOCK * q8 * 8 + bl * 8;
        for (size_t o = 0; o < q8; ++o) std::memcpy(masked + 8 * o, base + o * nb * 8, 2 * std::min<size_t>(8, M - 8 * o));
    }
    return HRX_OK;
}

void hrx_position_major_sizes(size_t B, size_t M, size_t D, size_t *records_u32, size_t *masked_u16) {
    if (records_u32) *records_u32 = (M + 3) / 4 * B * 4 * D;
    if (masked_u16) *masked_u16 = (M + 7) / 8 * B * 8;
}

void hrx_recommended_pitches(size_t M, size_t *rec_pitch, size_t *msk_pitch, size_t *chars_stride) {
    // Strides that are powers of two (4 KiB of records per string at M = 1024) put the chip-wide write front on a
    // fraction of the HBM channels: measured 4.3 vs 5.2 TB/s (tools/wpattern2).  One extra 128-byte line per string
    // breaks the pattern; masked rows keep 64-row (128-byte) alignment.
    const size_t m64 = (M + 63) / 64 * 64;
    if (rec_pitch) *rec_pitch = m64 + 32;
    if (msk_pitch) *msk_pitch = m64 + 64;
    if (chars_stride) *chars_stride = (M + 127) / 128 * 128 + 128;
}

}  // extern "C"

