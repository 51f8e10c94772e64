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
}

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
    /// what the context's last placement-aware allocation did (steps measured, reference / first / kept probe times, memory held)
    pub fn hrx_alloc_last_report(ctx: *const hrx_ctx, out: *mut hrx_place_report) -> c_int;
    /// roofline diagnostic: the memory traffic of one position-major launch over these buffers, no DFA work (overwrites the outputs)
    pub fn hrx_traffic_pass_device(ctx: *mut hrx_ctx, chars: *const u8, stride: usize, b: usize, m: usize, records: *mut u32,
                                   masked: *mut u16, stream: *mut c_void) -> c_int;
    /// device = HRX_DEVICE_NONE (-1): a host-only context (the native small-batch walk; no GPU needed)
    pub fn hrx_device_count(count: *mut c_int) -> c_int;
    pub fn hrx_ctx_device(ctx: *const hrx_ctx) -> c_int;
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
    fn drop(&mut self) { unsafe { hrx_ctx_destroy(self.ctx); hrx_defs_destroy(self.defs); } }
}
