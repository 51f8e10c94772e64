//! hrx.rs — the binding a halo2-regex maintainer would add (e.g. as `src/hrx.rs`) to route the witness path of
//! `RegexVerifyConfig::match_substrs` through libhrx.so.  NOT compiled in this repository (the image has no
//! Rust toolchain); it documents the drop-in call sites against include/hrx.h.  See INTEGRATION.md.
#![allow(non_camel_case_types)]
use std::ffi::{c_char, c_int, c_void, CStr};

#[repr(C)] pub struct hrx_defs { _p: [u8; 0] }
#[repr(C)] pub struct hrx_ctx { _p: [u8; 0] }
#[repr(C)] pub struct hrx_multi { _p: [u8; 0] }
#[repr(C)] #[derive(Default, Clone, Copy)]
pub struct hrx_place_report {
    pub searched: c_int, pub steps: c_int, pub accepted: c_int, pub chosen_step: c_int,
    pub ref_us: f64, pub first_us: f64, pub best_us: f64,
    pub ref_gbs: f64, pub first_gbs: f64, pub best_gbs: f64,
    pub probe_bytes: usize, pub peak_candidate_bytes: usize, pub search_ms: f64,
    pub capped: c_int,   // HRX_PLACE_CAPPED_* or-ed: which bound ended the walk (0: the acceptance rule itself)
}
pub const HRX_PLACE_OFF: c_int = 0;
pub const HRX_PLACE_WALK: c_int = 1;
pub const HRX_PLACE_CAPPED_STEPS: c_int = 1;
pub const HRX_PLACE_CAPPED_BYTES: c_int = 2;
pub const HRX_PLACE_CAPPED_TIME: c_int = 4;
pub const HRX_PLACE_CAPPED_ALLOC: c_int = 8;

pub const HRX_DEVICE_NONE: c_int = -1;
pub const HRX_MAX_DEFS: usize = 32;   // RegexDefs per config (more than three are walked in passes)

#[link(name = "hrx")]
extern "C" {
    pub fn hrx_defs_create(out: *mut *mut hrx_defs) -> c_int;
    pub fn hrx_defs_destroy(defs: *mut hrx_defs);
    pub fn hrx_defs_push_allstr(defs: *mut hrx_defs, first: u64, accepted: u64, largest: u64, n: usize,
                                cur: *const u64, next: *const u64, chr: *const u8, line_idx: *const u64) -> c_int;
    pub fn hrx_defs_push_substr(defs: *mut hrx_defs, n_pairs: usize, pair_cur: *const u64, pair_next: *const u64,
                                n_start: usize, starts: *const u64, n_end: usize, ends: *const u64) -> c_int;
    pub fn hrx_defs_finalize(defs: *mut hrx_defs) -> c_int;
    pub fn hrx_ctx_create(defs: *const hrx_defs, device: c_int, out: *mut *mut hrx_ctx) -> c_int;
    pub fn hrx_ctx_destroy(ctx: *mut hrx_ctx);
    pub fn hrx_last_error() -> *const c_char;
    pub fn hrx_witness_batch_host(ctx: *mut hrx_ctx, chars: *const u8, stride: usize, lens: *const u32, b: usize,
                                  m: usize, records: *mut u32, masked: *mut u16, status: *mut u64) -> c_int;
    pub fn hrx_witness_batch_device(ctx: *mut hrx_ctx, chars: *const u8, stride: usize, lens: *const u32, b: usize,
                                    m: usize, records: *mut u32, masked: *mut u16, status: *mut u64,
                                    stream: *mut c_void) -> c_int;
    /// multi-GPU driver for host buffers: shards by string index, one context and host thread per listed device
    pub fn hrx_multi_create(defs: *const hrx_defs, devices: *const c_int, n_devices: c_int, out: *mut *mut hrx_multi) -> c_int;
    pub fn hrx_multi_destroy(m: *mut hrx_multi);
    pub fn hrx_multi_witness_batch_host(m: *mut hrx_multi, chars: *const u8, stride: usize, lens: *const u32, b: usize, max_chars_size: usize,
                                        records: *mut u32, masked: *mut u16, status: *mut u64) -> c_int;
    /// device-resident shards: one set of DEVICE pointers per shard (on hrx_multi_shard_device(m, r)), one kernel per shard on
    /// its own stream, asynchronous; hrx_multi_synchronize waits.  No PCIe traffic, no collective.
    pub fn hrx_multi_num_shards(m: *const hrx_multi) -> c_int;
    pub fn hrx_multi_shard_device(m: *const hrx_multi, shard: c_int) -> c_int;
    /// the shard's private hipStream_t: make it wait on an event behind whatever produced the shard's inputs
    pub fn hrx_multi_shard_stream(m: *const hrx_multi, shard: c_int) -> *mut c_void;
    pub fn hrx_multi_witness_batch_device(m: *mut hrx_multi, layout: c_int, chars: *const *const u8, stride: usize, lens: *const *const u32,
                                          counts: *const usize, max_chars_size: usize, records: *const *mut u32, masked: *const *mut u16,
                                          status: *const *mut u64) -> c_int;
    pub fn hrx_multi_synchronize(m: *mut hrx_multi) -> c_int;
    pub fn hrx_shard_range(b: usize, world: c_int, rank: c_int, begin: *mut usize, count: *mut usize);
    /// position-major layouts (HRX_LAYOUT_POSITION_MAJOR = 1, | HRX_LAYOUT_INPUT_POSITION_MAJOR = 2): the coalesced fast path
    pub fn hrx_witness_batch_device_layout(ctx: *mut hrx_ctx, layout: c_int, chars: *const u8, stride: usize, lens: *const u32, b: usize,
                                           m: usize, records: *mut u32, masked: *mut u16, status: *mut u64, stream: *mut c_void) -> c_int;
    pub fn hrx_position_major_sizes(b: usize, m: usize, d: usize, n_records_u32: *mut usize, n_masked_u16: *mut usize);
    /// placement-aware allocation of the two output buffers of a large position-major batch; release each with hrx_device_free
    pub fn hrx_alloc_outputs_position_major(ctx: *mut hrx_ctx, b: usize, m: usize, records: *mut *mut u32, masked: *mut *mut u16) -> c_int;
    pub fn hrx_alloc_output_pair(ctx: *mut hrx_ctx, records_bytes: usize, masked_bytes: usize, records: *mut *mut c_void, masked: *mut *mut c_void) -> c_int;
    pub fn hrx_device_free(ptr: *mut c_void) -> c_int;
    /// B contiguous strings on the device (one &[u8] per string: lib.rs:311-315) -> HRX_LAYOUT_INPUT_POSITION_MAJOR, the fast path's input
    pub fn hrx_chars_to_position_major_device(ctx: *mut hrx_ctx, chars: *const u8, stride: usize, b: usize, chars_pm: *mut u8, stream: *mut c_void) -> c_int;
    /// what the context's last placement-aware allocation did (steps measured, reference / first / kept probe times, memory held)
    pub fn hrx_alloc_last_report(ctx: *const hrx_ctx, out: *mut hrx_place_report) -> c_int;
    /// placement per context: HRX_PLACE_OFF = two plain allocations; max_bytes / max_ms bound one walk (0 = the defaults)
    pub fn hrx_ctx_set_placement(ctx: *mut hrx_ctx, mode: c_int, max_bytes: usize, max_ms: f64) -> c_int;
    /// roofline diagnostic: the memory traffic of one position-major launch over these buffers, no DFA work (overwrites the outputs)
    pub fn hrx_traffic_pass_device(ctx: *mut hrx_ctx, chars: *const u8, stride: usize, b: usize, m: usize, records: *mut u32,
                                   masked: *mut u16, stream: *mut c_void) -> c_int;
    pub fn hrx_traffic_pass_device_layout(ctx: *mut hrx_ctx, layout: c_int, chars: *const u8, stride: usize, b: usize, m: usize, records: *mut u32, rec_pitch: usize,
                                          masked: *mut u16, msk_pitch: usize, stream: *mut c_void) -> c_int;
    /// one circuit's rows out of position-major HOST buffers: records [M][D], masked [M] (either pair may be null)
    pub fn hrx_rows_of_string_position_major(records_pm: *const u32, masked_pm: *const u16, b_total: usize, m: usize, d: usize, b: usize,
                                             records: *mut u32, masked: *mut u16) -> c_int;
    /// device = HRX_DEVICE_NONE (-1): a host-only context (the native small-batch walk; no GPU needed)
    pub fn hrx_device_count(count: *mut c_int) -> c_int;
    pub fn hrx_ctx_device(ctx: *const hrx_ctx) -> c_int;
    /// a second context of the same config (own stream / scratch / lock): what `impl Clone for RegexVerifyConfig` calls; device -2 = the source's
    pub fn hrx_ctx_clone(ctx: *const hrx_ctx, device: c_int, out: *mut *mut hrx_ctx) -> c_int;
    /// host-buffer batches of fewer than `rows` witness rows (B x M) take the native host walk (default 32768)
    pub fn hrx_ctx_set_host_threshold(ctx: *mut hrx_ctx, rows: usize) -> c_int;
    pub fn hrx_ctx_host_threshold(ctx: *const hrx_ctx) -> usize;
    /// match_substrs' integer columns for one string (lib.rs:311-773); any pointer may be null
    pub fn hrx_match_substrs(ctx: *mut hrx_ctx, characters: *const u8, n: usize, m: usize, enable: *mut u64, character: *mut u64,
                             state: *mut u64, substr_id: *mut u64, start_enable: *mut u64, end_enable: *mut u64,
                             masked_char: *mut u64, masked_substr_id: *mut u64, status: *mut u64) -> c_int;
    /// RegexTableConfig::load rows (table.rs:61-198): two-call pattern, returns the row count
    pub fn hrx_table_transition_rows(defs: *const hrx_defs, def: usize, rows4: *mut u64, cap_rows: usize) -> usize;
    pub fn hrx_table_endpoint_rows(defs: *const hrx_defs, def: usize, rows3: *mut u64, cap_rows: usize) -> usize;
    /// compact rows -> bn256::Fr cells ([4 + 4 D][b_count][M][4 x u64], Montgomery form; flags 1 = canonical)
    pub fn hrx_fr_columns_device(ctx: *mut hrx_ctx, layout: c_int, chars: *const u8, stride: usize, lens: *const u32,
                                 records: *const u32, rec_pitch: usize, masked: *const u16, msk_pitch: usize, b: usize, m: usize,
                                 b_begin: usize, b_count: usize, cells: *mut u64, flags: c_int, stream: *mut c_void) -> c_int;
    pub fn hrx_derive_states(ctx: *mut hrx_ctx, characters: *const u8, n: usize, states: *mut u64) -> c_int;
    pub fn hrx_derive_substr_ids(ctx: *mut hrx_ctx, states: *const u64, n: usize, substr_ids: *mut u64) -> c_int;
    pub fn hrx_derive_is_start_end(ctx: *mut hrx_ctx, states: *const u64, substr_ids: *const u64, n: usize,
                                   is_start: *mut u8, is_end: *mut u8) -> c_int;
    // definition generation (replaces the js_sandbox / fancy-regex path of src/vrm)
    pub fn hrx_gen_regex_files(parts: *const hrx_regex_part, n_parts: usize, max_byte_size: usize,
                               out: *mut *mut hrx_regex_files) -> c_int;
    pub fn hrx_regex_files_num_substrs(files: *const hrx_regex_files) -> usize;
    pub fn hrx_regex_files_allstr(files: *const hrx_regex_files, len: *mut usize) -> *const c_char;
    pub fn hrx_regex_files_substr(files: *const hrx_regex_files, idx: usize, len: *mut usize) -> *const c_char;
    pub fn hrx_regex_files_destroy(files: *mut hrx_regex_files);
    // ---- the rest of include/hrx.h (tests/test_abi.py diffs these names against the header) ----
    /// the text parsers of src/defs.rs (read_from_text / read_from_reader) for callers that hold files or strings instead of the structs
    pub fn hrx_defs_push_allstr_text(defs: *mut hrx_defs, text: *const c_char, len: usize) -> c_int;
    pub fn hrx_defs_push_allstr_file(defs: *mut hrx_defs, path: *const c_char) -> c_int;
    pub fn hrx_defs_push_substr_text(defs: *mut hrx_defs, text: *const c_char, len: usize) -> c_int;
    pub fn hrx_defs_push_substr_file(defs: *mut hrx_defs, path: *const c_char) -> c_int;
    pub fn hrx_defs_num_defs(defs: *const hrx_defs) -> usize;
    pub fn hrx_defs_num_substrs(defs: *const hrx_defs, def: usize) -> usize;
    pub fn hrx_defs_first_state(defs: *const hrx_defs, def: usize) -> u64;
    pub fn hrx_defs_accepted_state(defs: *const hrx_defs, def: usize) -> u64;
    pub fn hrx_defs_largest_state(defs: *const hrx_defs, def: usize) -> u64;
    pub fn hrx_defs_num_transitions(defs: *const hrx_defs, def: usize) -> usize;
    pub fn hrx_defs_substr_id_offset(defs: *const hrx_defs, def: usize) -> u64;
    pub fn hrx_defs_table_bytes(defs: *const hrx_defs) -> usize;
    pub fn hrx_witness_batch_device_pitched(ctx: *mut hrx_ctx, chars: *const u8, stride: usize, lens: *const u32, b: usize, m: usize, records: *mut u32, rec_pitch: usize,
                                            masked: *mut u16, msk_pitch: usize, status: *mut u64, stream: *mut c_void) -> c_int;
    pub fn hrx_recommended_pitches(m: usize, rec_pitch: *mut usize, msk_pitch: *mut usize, chars_stride: *mut usize);
    /// RECORD PLANES: every def's records in a buffer of its own (the D + 1 write streams of a launch spread over the classes of the device memory)
    pub fn hrx_witness_batch_device_planes(ctx: *mut hrx_ctx, layout: c_int, chars: *const u8, stride: usize, lens: *const u32, b: usize, m: usize,
                                           record_planes: *const *mut u32, n_planes: usize, masked: *mut u16, status: *mut u64, stream: *mut c_void) -> c_int;
    pub fn hrx_probe_write_pair(ctx: *mut hrx_ctx, a: *mut c_void, b: *mut c_void, bytes: usize, gbs: *mut f64) -> c_int;
    pub fn hrx_position_major_plane_sizes(b: usize, m: usize, plane_u32: *mut usize, masked_u16: *mut usize);
    pub fn hrx_position_major_stripe_sizes(b: usize, m: usize, n_stripes: usize, stripe_u32: *mut usize, masked_u16: *mut usize);
    pub fn hrx_alloc_output_planes(ctx: *mut hrx_ctx, b: usize, m: usize, n_planes: usize, record_planes: *mut *mut u32, masked: *mut *mut u16) -> c_int;
    pub fn hrx_alloc_output_planes_for_batch(ctx: *mut hrx_ctx, layout: c_int, chars: *const u8, stride: usize, lens: *const u32, b: usize, m: usize, n_planes: usize,
                                             record_planes: *mut *mut u32, masked: *mut *mut u16) -> c_int;
    pub fn hrx_rows_of_string_planes(record_planes: *const *const u32, n_planes: usize, masked_pm: *const u16, b_total: usize, m: usize, d: usize, b: usize,
                                     records: *mut u32, masked: *mut u16) -> c_int;
    pub fn hrx_traffic_pass_device_planes(ctx: *mut hrx_ctx, chars: *const u8, stride: usize, b: usize, m: usize, record_planes: *const *mut u32, n_planes: usize,
                                          masked: *mut u16, stream: *mut c_void) -> c_int;
    /// per-context choices between variants that compute the same rows (HRX_OPT_*)
    pub fn hrx_ctx_host_route_report(ctx: *const hrx_ctx, out: *mut hrx_host_route_report, out_bytes: usize) -> c_int;
    pub fn hrx_alloc_last_report_sized(ctx: *const hrx_ctx, out: *mut c_void, out_bytes: usize) -> c_int;
    pub fn hrx_ctx_set_option(ctx: *mut hrx_ctx, option: c_int, value: std::ffi::c_long) -> c_int;
    pub fn hrx_ctx_get_option(ctx: *const hrx_ctx, option: c_int) -> std::ffi::c_long;
    pub fn hrx_describe_launch(defs: *const hrx_defs, layout: c_int, b: usize, m: usize, num_cus: c_int, out: *mut c_char, cap: usize) -> c_int;
    pub fn hrx_ctx_describe_launch(ctx: *const hrx_ctx, layout: c_int, b: usize, m: usize, out: *mut c_char, cap: usize) -> c_int;
    /// the Vecs of lib.rs:316-318 out of one string's compact records; the integer advice columns of all circuits of a batch, column-major
    pub fn hrx_witness_of_string(records: *const u32, d: usize, n: usize, m: usize, states: *mut u64, substr_ids: *mut usize, is_start: *mut bool, is_end: *mut bool) -> c_int;
    pub fn hrx_witness_num_columns(d: usize) -> usize;
    pub fn hrx_witness_columns_host(layout: c_int, chars: *const u8, stride: usize, lens: *const u32, records: *const u32, rec_pitch: usize, masked: *const u16,
                                    msk_pitch: usize, b: usize, m: usize, d: usize, b_begin: usize, b_count: usize, columns: *mut u64) -> c_int;
    pub fn hrx_fr_columns_device_planes(ctx: *mut hrx_ctx, layout: c_int, chars: *const u8, stride: usize, lens: *const u32, record_planes: *const *const u32, n_planes: usize,
                                        masked: *const u16, b: usize, m: usize, b_begin: usize, b_count: usize, cells: *mut u64, flags: c_int, stream: *mut c_void) -> c_int;
    pub fn hrx_fr_num_columns(d: usize) -> usize;
    pub fn hrx_fr_from_u64(v: u64, flags: c_int, limbs: *mut u64);
    /// what src/vrm/js_caller.rs:36-48, 127-157 obtain from V8: regexToDfa's JSON, the AllstrRegexDef text, formatRegexPrintable; and the part search of vrm/mod.rs:540-600
    pub fn hrx_regex_to_allstr_text(regex: *const c_char, regex_len: usize, out: *mut c_char, cap: usize, needed: *mut usize) -> c_int;
    pub fn hrx_regex_to_dfa_json(regex: *const c_char, regex_len: usize, out: *mut c_char, cap: usize, needed: *mut usize) -> c_int;
    pub fn hrx_format_regex_str(regex: *const c_char, regex_len: usize, out: *mut c_char, cap: usize, needed: *mut usize) -> c_int;
    pub fn hrx_regex_find(pattern: *const c_char, pattern_len: usize, text: *const c_char, text_len: usize, found: *mut c_int, start: *mut usize, end: *mut usize) -> c_int;
}
pub const HRX_OPT_PMD_COMBINER_WAVE: c_int = 1;
pub const HRX_OPT_HOST_ROUTE: c_int = 2;      // HRX_HOST_ROUTE_AUTO (the fastest of device / host cores / both at once, by measurement) / _DEVICE / _HOST
pub const HRX_OPT_HOST_THREADS: c_int = 3;
pub const HRX_OPT_HOST_PIPELINE: c_int = 4;
pub const HRX_OPT_PLACE_DRY_LAUNCH: c_int = 5;
#[repr(C)] #[derive(Default, Clone, Copy)]
pub struct hrx_host_route_report {
    pub route: c_int, pub device_strings: usize, pub host_strings: usize, pub device_ms: f64, pub host_ms: f64, pub call_ms: f64,
    pub device_alone_ns_per_row: f64, pub host_alone_ns_per_row: f64, pub split_ns_per_row: f64,
    pub device_ns_per_row: f64, pub host_ns_per_row: f64, pub host_threads: c_int, pub device_pipelined: c_int,
}

#[repr(C)] pub struct hrx_regex_part { regex_def: *const c_char, regex_len: usize, is_public: c_int, max_size: usize }
#[repr(C)] pub struct hrx_regex_files { _private: [u8; 0] }

/// Drop-in body for `DecomposedRegexConfig::gen_regex_files` (src/vrm/mod.rs:62-307): same files, no V8.
pub fn gen_regex_files(cfg: &crate::vrm::DecomposedRegexConfig, allstr_file_path: &std::path::PathBuf,
                       substr_file_pathes: &[std::path::PathBuf]) -> Result<(), crate::vrm::VrmError> {
    let parts: Vec<hrx_regex_part> = cfg.parts.iter().map(|p| hrx_regex_part {
        regex_def: p.regex_def.as_ptr() as *const c_char, regex_len: p.regex_def.len(),
        is_public: p.is_public as c_int, max_size: p.max_size }).collect();
    unsafe {
        let mut files = std::ptr::null_mut();
        if hrx_gen_regex_files(parts.as_ptr(), parts.len(), cfg.max_byte_size, &mut files) != 0 { panic!("{}", last_error()); }
        let text = |p: *const c_char, n: usize| std::slice::from_raw_parts(p as *const u8, n).to_vec();
        let mut n = 0usize;
        let p = hrx_regex_files_allstr(files, &mut n);
        std::fs::write(allstr_file_path, text(p, n))?;
        for idx in 0..hrx_regex_files_num_substrs(files) {
            let p = hrx_regex_files_substr(files, idx, &mut n);
            std::fs::write(&substr_file_pathes[idx], text(p, n))?;
        }
        hrx_regex_files_destroy(files);
    }
    Ok(())
}

fn last_error() -> String { unsafe { CStr::from_ptr(hrx_last_error()).to_string_lossy().into_owned() } }

/// Owns the device context built from `Vec<RegexDefs>`; shared (Arc) by the clones halo2 makes of the config.
pub struct HrxHandle { defs: *mut hrx_defs, ctx: *mut hrx_ctx }
unsafe impl Send for HrxHandle {}
unsafe impl Sync for HrxHandle {} // calls on one ctx are serialised inside the library

impl HrxHandle {
    /// From the structs of src/defs.rs: state_lookup entries in any order with their line indices (table.rs:108 order).
    pub fn new(regex_defs: &[crate::defs::RegexDefs], device: i32) -> Self {
        unsafe {
            let mut defs = std::ptr::null_mut();
            assert_eq!(hrx_defs_create(&mut defs), 0);
            for rd in regex_defs {
                let (mut cur, mut next, mut chr, mut idx) = (vec![], vec![], vec![], vec![]);
                for ((c, s), (i, n)) in rd.allstr.state_lookup.iter() { cur.push(*s); next.push(*n); chr.push(*c); idx.push(*i as u64); }
                assert_eq!(hrx_defs_push_allstr(defs, rd.allstr.first_state_val, rd.allstr.accepted_state_val,
                    rd.allstr.largest_state_val, cur.len(), cur.as_ptr(), next.as_ptr(), chr.as_ptr(), idx.as_ptr()), 0, "{}", last_error());
                for sd in rd.substrs.iter() {
                    let (a, b): (Vec<u64>, Vec<u64>) = sd.valid_state_transitions.iter().cloned().unzip();
                    assert_eq!(hrx_defs_push_substr(defs, a.len(), a.as_ptr(), b.as_ptr(), sd.start_states.len(),
                        sd.start_states.as_ptr(), sd.end_states.len(), sd.end_states.as_ptr()), 0, "{}", last_error());
                }
            }
            assert_eq!(hrx_defs_finalize(defs), 0, "{}", last_error());
            let mut ctx = std::ptr::null_mut();
            assert_eq!(hrx_ctx_create(defs, device, &mut ctx), 0, "{}", last_error());
            HrxHandle { defs, ctx }
        }
    }

    /// Drop-in for `RegexVerifyConfig::derive_states` (src/lib.rs:804-823), same panic text.
    pub fn derive_states(&self, num_defs: usize, characters: &[u8]) -> Vec<Vec<u64>> {
        let n = characters.len();
        let mut flat = vec![0u64; num_defs * (n + 1)];
        let rc = unsafe { hrx_derive_states(self.ctx, characters.as_ptr(), n, flat.as_mut_ptr()) };
        if rc != 0 { panic!("{}", last_error()); } // "The transition from {} by {} is invalid!"
        flat.chunks(n + 1).map(|c| c.to_vec()).collect()
    }

    /// Drop-in for `RegexVerifyConfig::derive_substr_ids` (src/lib.rs:825-845): `states` as derive_states returned them.
    pub fn derive_substr_ids(&self, states: &[Vec<u64>]) -> Vec<Vec<usize>> {
        let n = states[0].len() - 1;
        let flat: Vec<u64> = states.iter().flat_map(|s| s.iter().cloned()).collect();
        let mut out = vec![0u64; states.len() * n];
        let rc = unsafe { hrx_derive_substr_ids(self.ctx, flat.as_ptr(), n, out.as_mut_ptr()) };
        if rc != 0 { panic!("{}", last_error()); }
        out.chunks(n.max(1)).take(states.len()).map(|c| c.iter().map(|v| *v as usize).collect()).collect()
    }

    /// Drop-in for `RegexVerifyConfig::derive_is_start_end` (src/lib.rs:847-888): (is_starts, is_ends), n + 1 flags per def each
    /// (is_starts[d][n] = false, is_ends[d][0] = false, is_ends[d][i + 1] belongs to transition i).
    pub fn derive_is_start_end(&self, states: &[Vec<u64>], substr_ids: &[Vec<usize>]) -> (Vec<Vec<bool>>, Vec<Vec<bool>>) {
        let n = states[0].len() - 1;
        let d = states.len();
        let fs: Vec<u64> = states.iter().flat_map(|s| s.iter().cloned()).collect();
        let fi: Vec<u64> = substr_ids.iter().flat_map(|s| s.iter().map(|v| *v as u64)).collect();
        let (mut st, mut en) = (vec![0u8; d * (n + 1)], vec![0u8; d * (n + 1)]);
        let rc = unsafe { hrx_derive_is_start_end(self.ctx, fs.as_ptr(), fi.as_ptr(), n, st.as_mut_ptr(), en.as_mut_ptr()) };
        if rc != 0 { panic!("{}", last_error()); }
        let split = |v: &Vec<u8>| v.chunks(n + 1).map(|c| c.iter().map(|x| *x != 0).collect()).collect();
        (split(&st), split(&en))
    }

    /// `match_substrs`' integer columns for ONE string (src/lib.rs:311-773 on integers): what the cells of a circuit hold, in the order
    /// enable, character, states[d], substr_ids[d], start_enable[d], end_enable[d], masked_char, masked_substr_id.
    pub fn match_substrs_columns(&self, num_defs: usize, characters: &[u8], max_chars_size: usize) -> MatchColumns {
        let (m, d) = (max_chars_size, num_defs);
        let mut c = MatchColumns { enable: vec![0; m], character: vec![0; m], state: vec![0; d * m], substr_id: vec![0; d * m],
                                   start_enable: vec![0; d * m], end_enable: vec![0; d * m], masked_char: vec![0; m], masked_substr_id: vec![0; m], status: 0 };
        let rc = unsafe { hrx_match_substrs(self.ctx, characters.as_ptr(), characters.len(), m, c.enable.as_mut_ptr(), c.character.as_mut_ptr(),
                                            c.state.as_mut_ptr(), c.substr_id.as_mut_ptr(), c.start_enable.as_mut_ptr(), c.end_enable.as_mut_ptr(),
                                            c.masked_char.as_mut_ptr(), c.masked_substr_id.as_mut_ptr(), &mut c.status) };
        if rc != 0 { panic!("{}", last_error()); }
        c
    }

    /// The batch surface: compact witness rows for many strings (one circuit each) sharing this config.
    /// records[b][r][d] = state | substr_id<<16 | start_enable<<24 | end_enable<<25; masked[b][r] = char | id<<8.
    pub fn witness_batch(&self, num_defs: usize, chars: &[u8], stride: usize, lens: &[u32], max_chars_size: usize)
        -> (Vec<u32>, Vec<u16>, Vec<u64>) {
        let b = lens.len();
        let (mut rec, mut msk, mut st) = (vec![0u32; b * max_chars_size * num_defs], vec![0u16; b * max_chars_size], vec![0u64; b]);
        let rc = unsafe { hrx_witness_batch_host(self.ctx, chars.as_ptr(), stride, lens.as_ptr(), b, max_chars_size,
                                                 rec.as_mut_ptr(), msk.as_mut_ptr(), st.as_mut_ptr()) };
        if rc != 0 { panic!("{}", last_error()); }
        for (i, s) in st.iter().enumerate() {
            if s & 0xff == 1 { panic!("string {}: The transition from {} by {} is invalid!", i, (s >> 24) & 0xffff, (s >> 16) & 0xff); }
        }
        (rec, msk, st)
    }
}

impl Drop for HrxHandle {
    fn drop(&mut self) { unsafe { hrx_ctx_destroy(self.ctx); if !self.defs.is_null() { hrx_defs_destroy(self.defs); } } }
}

impl HrxHandle {
    /// A second context of the same config on the same device — own stream, scratch and lock (hrx_ctx_clone): what `impl Clone for RegexVerifyConfig` calls when the
    /// prover's threads should overlap on the device instead of sharing one handle behind its mutex.  The clone needs neither the source nor its hrx_defs afterwards.
    pub fn clone_ctx(&self) -> HrxHandle {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { hrx_ctx_clone(self.ctx, -2 /* HRX_DEVICE_SAME */, &mut ctx) };
        assert_eq!(rc, 0, "{}", last_error());
        HrxHandle { defs: std::ptr::null_mut(), ctx }
    }

    /// One circuit's rows out of position-major HOST buffers (the device buffers copied out as they are): records [m][d], masked [m].
    pub fn rows_of_string(records_pm: &[u32], masked_pm: &[u16], batch: usize, m: usize, d: usize, b: usize) -> (Vec<u32>, Vec<u16>) {
        let (mut rec, mut msk) = (vec![0u32; m * d], vec![0u16; m]);
        let rc = unsafe { hrx_rows_of_string_position_major(records_pm.as_ptr(), masked_pm.as_ptr(), batch, m, d, b, rec.as_mut_ptr(), msk.as_mut_ptr()) };
        assert_eq!(rc, 0, "{}", last_error());
        (rec, msk)
    }
}

/// The integer columns of one circuit (hrx_match_substrs).  state / substr_id / start_enable / end_enable are [def][row].
pub struct MatchColumns {
    pub enable: Vec<u64>, pub character: Vec<u64>, pub state: Vec<u64>, pub substr_id: Vec<u64>, pub start_enable: Vec<u64>, pub end_enable: Vec<u64>,
    pub masked_char: Vec<u64>, pub masked_substr_id: Vec<u64>, pub status: u64,
}

// ---------------------------------------------------------------------------------------------------------------------
// SURVEY §8 f3 — the batch-aware witness fill (src/lib.rs:339-773), as far as it can be written without a Rust toolchain.
//
// `match_substrs` starts with three lines (lib.rs:316-318):
//     let states = self.derive_states(characters);
//     let substr_ids = self.derive_substr_ids(states.as_slice());
//     let (is_starts, is_ends) = self.derive_is_start_end(&states, &substr_ids);
// and everything below them (lib.rs:339-773: the advice assignments, the accept chain, the per-def sums and the reveal-mask gates) reads ONLY those four
// values, `characters` and `self`.  So the fill of circuit b of a batch is the same body fed from the compact records of string b:
//     1. move the body below lib.rs:318 into `fn match_substrs_with(&self, ctx, characters, states, substr_ids, is_starts, is_ends)`
//        (a pure refactor: `match_substrs` = the three lines + that call);
//     2. per circuit of the batch: `let w = WitnessOf::new(&records[b * m * d ..][.. m * d], d, characters.len(), &config.regex_defs);`
//        `config.match_substrs_with(ctx, characters, &w.states, &w.substr_ids, &w.is_starts, &w.is_ends)`.
// The decode below is exact, not approximate: for idx < n the enable cell is 1, so start_enable[idx] = is_start[idx] and end_enable[idx] = is_end[idx + 1]
// (lib.rs:467-519); is_starts[n] is false by construction (lib.rs:868), is_ends[0] likewise (lib.rs:881), and is_ends[n] is the flag of transition
// n - 1 = end_enable[n - 1].  The masked columns the kernels also return are NOT needed by the fill (the gates recompute them, lib.rs:593-764); a prover that
// wants to skip witness generation inside the gates can compare them against `masked`.
// ---------------------------------------------------------------------------------------------------------------------
pub struct WitnessOf {
    pub states: Vec<Vec<u64>>,        // [def][n + 1]   as derive_states returns them
    pub substr_ids: Vec<Vec<usize>>,  // [def][n]       as derive_substr_ids
    pub is_starts: Vec<Vec<bool>>,    // [def][n + 1]   as derive_is_start_end
    pub is_ends: Vec<Vec<bool>>,      // [def][n + 1]
}

impl WitnessOf {
    /// `rec`: the m x d compact records of one string, string-major (record = state | substr_id << 16 | start_enable << 24 | end_enable << 25);
    /// `n` = characters.len() <= m.  Row n of the records holds the state after the last character (lib.rs:404-411); when n == m that row does not
    /// exist and the state after the last character is not part of any cell either (the reference's states[d][m] is computed and never assigned).
    pub fn new(rec: &[u32], d: usize, n: usize, m: usize) -> Self {
        assert!(rec.len() >= m * d && n <= m);
        // hrx_witness_of_string (csrc/hrx_fill.cpp) writes [def][..] row-major; Vec<bool> is one 0 / 1 byte per element
        let (mut states, mut ids) = (vec![0u64; d * (n + 1)], vec![0usize; d * n]);
        let (mut st, mut en) = (vec![false; d * (n + 1)], vec![false; d * (n + 1)]);
        let rc = unsafe { hrx_witness_of_string(rec.as_ptr(), d, n, m, states.as_mut_ptr(), ids.as_mut_ptr(), st.as_mut_ptr(), en.as_mut_ptr()) };
        if rc != 0 { panic!("{}", last_error()); }
        WitnessOf {
            states: states.chunks(n + 1).map(|c| c.to_vec()).collect(),
            substr_ids: if n == 0 { vec![vec![]; d] } else { ids.chunks(n).map(|c| c.to_vec()).collect() },
            is_starts: st.chunks(n + 1).map(|c| c.to_vec()).collect(),
            is_ends: en.chunks(n + 1).map(|c| c.to_vec()).collect(),
        }
        // (n == m: states[di][m] is 0 and is_ends[di][m] false — the two values match_substrs computes and never assigns: lib.rs:388-418, 501)
    }

    /// The same view out of HRX_LAYOUT_POSITION_MAJOR records ([ceil(m/4)][d][nb][4] per block of 65536 strings): string `b` of a batch of `batch` strings.
    pub fn from_position_major(rec_pm: &[u32], batch: usize, b: usize, d: usize, n: usize, m: usize) -> Self {
        let mut flat = vec![0u32; m * d];
        let rc = unsafe { hrx_rows_of_string_position_major(rec_pm.as_ptr(), std::ptr::null(), batch, m, d, b, flat.as_mut_ptr(), std::ptr::null_mut()) };
        if rc != 0 { panic!("{}", last_error()); }
        Self::new(&flat, d, n, m)
    }
}

