// hrx_describe_api.cpp — hrx_describe_launch / hrx_ctx_describe_launch: the kernel and geometry the planner picks for a shape, as text, without launching
// (what rocprofv3 will list; tests/test_abi.py pins it per config).
#include "hrx_ctx.hpp"
#include "hrx_lane.h"

using namespace hrx;

extern "C" {

static int describe_set(const DefsSet &s, int layout, size_t B, size_t M, int num_cus, std::string &out, bool summary_pass, const uint32_t dbg, const uint32_t tune) {
    WitnessArgs a{};
    a.layout = (uint32_t)layout; a.B = (uint32_t)B; a.M = (uint32_t)M;
    // the planner only looks at which images exist and how large they are
    a.table_image = s.table_image.data(); a.table_bytes = (uint32_t)(s.table_image.size() * 4);
    a.wide_image = s.wide_image.empty() ? nullptr : s.wide_image.data();
    a.half_image = s.half_image.empty() ? nullptr : s.half_image.data();
    a.half_bytes = (uint32_t)(s.half_image.size() * 2);
    a.pair_image = s.pair.image.empty() ? nullptr : s.pair.image.data(); a.pair_bytes = s.pair.bytes; a.pair_classes = s.pair.n_classes;
    a.pair_blk_bytes = s.pair.blk_bytes; a.pair_lut_off = s.pair.lut_off;
    a.byte_image = s.byte.image.empty() ? nullptr : s.byte.image.data(); a.byte_bytes = s.byte.bytes; a.byte_dead = s.byte.dead; a.byte16_bytes = s.byte.bytes16;
    a.D = (uint32_t)s.defs.size();
    a.debug = dbg;   // hrx_describe_launch: what a context created now would run with (kernel-selection bits only in a release build); hrx_ctx_describe_launch: the context's
    a.tune = tune;
    if (layout & HRX_LAYOUT_RECORD_PLANES) {     // the launch hrx_witness_batch_device_planes makes (launch_batch)
        a.layout &= ~(uint32_t)HRX_LAYOUT_RECORD_PLANES;
        layout &= ~HRX_LAYOUT_RECORD_PLANES;
        a.rec_planes[0] = reinterpret_cast<unsigned char *>(16);
        if (a.D == 1) { a.rec_stripes = 2; a.debug |= kDbgNoPair | kDbgNoSpec; }      // (one def: described with its two row stripes)
    }
    if (summary_pass) a.debug |= kDbgNoPair | kDbgNoDefParallel;   // (launch_batch: a pass of a multi-pass config is the loader / walker / finisher kernel)
    LaunchInfo li;
    if (!plan_witness_launch(a, num_cus, li)) return fail(HRX_ERR_BOUNDS, "tables + staging do not fit the 160 KiB LDS");
    if (summary_pass && li.half) {
        a.debug |= kDbgForceGlobalTable;
        if (!plan_witness_launch(a, num_cus, li)) return fail(HRX_ERR_BOUNDS, "tables + staging do not fit the 160 KiB LDS");
    }
    char name[128], line[256];
    const char *tf[2] = {"false", "true"};
    // the names rocprofv3 lists: every template argument spelled out, defaulted ones too (an exact-match join with a kernel_stats.csv works)
    if (li.split == 6) std::snprintf(name, sizeof name, "hrx::witness_pp_kernel");
    else if (li.split == 5) std::snprintf(name, sizeof name, "hrx::witness_pmd_kernel<%u, %s>", a.D, a.cw_image ? ((a.layout & 1u) ? "true, true, false" : "true, true, true") : li.pmd_fin ? "false, true, false" : "false, false, false");
    else if (li.split == 2) std::snprintf(name, sizeof name, "hrx::witness_pm_kernel<%u, %s, %s, %s, %s, %s>", a.D, tf[li.gtab], tf[li.wide], tf[li.half], tf[!(layout & 1)], tf[li.byte]);
    else if (li.split == 1) std::snprintf(name, sizeof name, "hrx::witness_split_kernel<%u, %u, %s>", a.D, li.byte ? 32u : 32u / a.D, tf[li.byte]);
    else std::snprintf(name, sizeof name, "hrx::witness_kernel<%u, %s, %s>", a.D, tf[(M % 8) == 0], tf[li.gtab]);
    std::snprintf(line, sizeof line, "%s grid=%d waves=%d ring=%d lds=%zu%s", name, li.grid, li.waves_per_wg, li.nslots, li.lds_bytes, li.dyn ? " groups=dynamic" : "");
    out = line;
    if (li.spec_tiles) {
        std::snprintf(line, sizeof line, " chunked=%dx%d tiles: hrx::spec_scout_kernel + hrx::spec_compose_kernel before, hrx::spec_stitch_kernel behind", li.spec_chunks, li.spec_tiles);
        out += line;
    }
    return HRX_OK;
}

static int describe_config(const DefsSet &s, const uint32_t dbg, const uint32_t tune, const bool mpc_on, int layout, size_t B, size_t M, int num_cus, char *out, size_t cap) {
    std::string text;
    if (s.groups.empty()) {
        const bool byte_split = !s.byte.image.empty() && !(dbg & (kDbgNoByte | kDbgForceHalf));
        const bool via_tp = layout == HRX_LAYOUT_STRING_MAJOR && M % 8 == 0 && !byte_split &&
                            (!s.byte.image.empty() || !s.half_image.empty()) && s.table_image.size() * 4 + wave_stage_bytes((int)s.defs.size(), 16) > kLdsLimit;
        const int rc = describe_set(s, via_tp ? HRX_LAYOUT_POSITION_MAJOR : layout, B, M, num_cus, text, false, dbg, tune);
        if (rc != HRX_OK) return rc;
        if (via_tp) text += " + hrx::transpose_pm_to_sm_kernel";
    } else if ([&] {   // four and five defs, string-major rows in multiples of 16: the def-parallel launch writes them itself
                   if (s.cw_image.empty() || mpc_on || layout != HRX_LAYOUT_STRING_MAJOR) return false;
                   WitnessArgs a{};
                   a.layout = HRX_LAYOUT_STRING_MAJOR; a.B = (uint32_t)B; a.M = (uint32_t)M; a.D = (uint32_t)s.defs.size();
                   a.debug = dbg;
                   a.cw_image = s.cw_image.data(); a.table_bytes = (uint32_t)s.cw_image.size();
                   LaunchInfo li{};
                   if (!plan_pmd_cw_sm(a, num_cus, li)) return false;
                   char buf[256];
                   std::snprintf(buf, sizeof buf, "hrx::witness_pmd_kernel<%u, true, true, true> grid=%d waves=%d ring=%d sub-tiles=%u lds=%zu", a.D, li.grid, li.waves_per_wg, li.nslots, a.sm_bufs, li.lds_bytes);
                   text = buf;
                   return true;
               }()) {
    } else if ([&] {   // 6 or 7 defs with CLASS-WIDE tables: one def-parallel launch over the whole config (position-major; string-major rows in multiples of 8 through the transposer)
                   const bool tp = !(layout & 1) && M % 8 == 0;
                   if (s.cw_image.empty() || mpc_on || !((layout & 1) || tp)) return false;
                   WitnessArgs a{};
                   a.layout = HRX_LAYOUT_POSITION_MAJOR | (layout & HRX_LAYOUT_INPUT_POSITION_MAJOR); a.B = (uint32_t)B; a.M = (uint32_t)M; a.D = (uint32_t)s.defs.size();
                   a.debug = dbg;
                   a.cw_image = s.cw_image.data(); a.table_bytes = (uint32_t)s.cw_image.size();
                   LaunchInfo li{};
                   if (!plan_pmd_cw(a, num_cus, li)) return false;
                   char buf[256];
                   std::snprintf(buf, sizeof buf, "hrx::witness_pmd_kernel<%u, true, true, false> grid=%d waves=%d ring=%d lds=%zu", a.D, li.grid, li.waves_per_wg, li.nslots, li.lds_bytes);
                   text = buf;
                   if (tp) text += " + hrx::transpose_pm_to_sm_kernel";
                   return true;
               }()) {
    } else if ([&] {   // more than eight defs with CW groups: one def-parallel launch per group of 4 .. 8 defs, then the combine launch
                   const bool tp = !(layout & 1) && M % 8 == 0;
                   if (s.cw_groups.empty() || !((layout & 1) || tp) || (dbg & kDbgNoDefParallel)) return false;
                   std::string t2 = "multi-pass, " + std::to_string(s.cw_groups.size()) + " groups: ";
                   for (size_t g = 0; g < s.cw_groups.size(); ++g) {
                       WitnessArgs a{};
                       a.layout = HRX_LAYOUT_POSITION_MAJOR | (layout & HRX_LAYOUT_INPUT_POSITION_MAJOR); a.B = (uint32_t)B; a.M = (uint32_t)M; a.D = (uint32_t)s.cw_groups[g].defs.size();
                       a.cw_image = s.cw_groups[g].cw_image.data(); a.table_bytes = (uint32_t)s.cw_groups[g].cw_image.size();
                       LaunchInfo li{};
                       if (!plan_pmd_cw(a, num_cus, li)) return false;
                       char buf[256];
                       std::snprintf(buf, sizeof buf, "[defs %u..%zu: hrx::witness_pmd_kernel<%u, true, true, false> grid=%d waves=%d ring=%d lds=%zu] ", s.cw_group_first[g],
                                     s.cw_group_first[g] + s.cw_groups[g].defs.size() - 1, a.D, li.grid, li.waves_per_wg, li.nslots, li.lds_bytes);
                       t2 += buf;
                   }
                   text = t2 + "+ hrx::witness_combine_summary_kernel";
                   if (tp) text += " + hrx::transpose_pm_to_sm_kernel";
                   return true;
               }()) {
    } else {   // one launch per group of defs (position-major, the caller's input layout), then the combine kernel
        text = "multi-pass, " + std::to_string(s.groups.size()) + " groups: ";
        for (size_t g = 0; g < s.groups.size(); ++g) {
            std::string one;
            const bool tp = !(layout & 1) && M % 8 == 0;
            const int rc = describe_set(s.groups[g], HRX_LAYOUT_POSITION_MAJOR | (layout & HRX_LAYOUT_INPUT_POSITION_MAJOR), B, M, num_cus, one, (layout & 1) != 0 || tp, dbg, tune);
            if (rc != HRX_OK) return rc;
            text += "[defs " + std::to_string(s.group_first[g]) + ".." + std::to_string(s.group_first[g] + s.groups[g].defs.size() - 1) + ": " + one + "] ";
        }
        const bool via_tp = !(layout & 1) && M % 8 == 0;
        const bool merge_last = ((layout & 1) || via_tp) && s.groups.size() - 1 <= kMaxMergeGroups && !mpc_on;
        text += merge_last ? "(the last pass merges the summaries) + hrx::witness_merge_status_kernel"
                           : (layout & 1) || via_tp ? "+ hrx::witness_combine_summary_kernel" : "+ hrx::witness_combine_kernel<true>";
        if (via_tp) text += " + hrx::transpose_pm_to_sm_kernel";
    }
    std::snprintf(out, cap, "%s", text.c_str());
    return HRX_OK;
}

int hrx_describe_launch(const hrx_defs *defs, int layout, size_t B, size_t M, int num_cus, char *out, size_t cap) {
    if (!defs || !out || !cap) return fail(HRX_ERR_ARG, "NULL argument");
    if (!defs->s.finalized) return fail(HRX_ERR_STATE, "call hrx_defs_finalize first");
    if (num_cus < 1) return fail(HRX_ERR_ARG, "num_cus must be >= 1");
    const char *mpc = std::getenv("HRX_MP_COMBINE");      // (what hrx_ctx_create would read now)
    return describe_config(defs->s, debug_flags_from_env(), 0u, mpc && std::atoi(mpc) != 0, layout, B, M, num_cus, out, cap);
}

int hrx_ctx_describe_launch(const hrx_ctx *ctx, int layout, size_t B, size_t M, char *out, size_t cap) {
    if (!ctx || !out || !cap) return fail(HRX_ERR_ARG, "NULL argument");
    return describe_config(ctx->s, ctx->debug, ctx->tune, ctx->mp_combine, layout, B, M, ctx->num_cus > 0 ? ctx->num_cus : 256, out, cap);
}

}  // extern "C"
