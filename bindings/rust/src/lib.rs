//! Rust host side of the C ABI in `include/tlamc.h` — the "Rust host over a thin C-ABI FFI" that
//! `BASELINE.json:north_star` asks for (the reference's README.md:12-16 planned a Rust implementation
//! checked with quickcheck).  NOT COMPILED IN THIS REPOSITORY'S CI: the build image has no Rust toolchain;
//! this file is the literal transcription of the header a maintainer would start from.
#![allow(non_camel_case_types)]
use std::ffi::{CStr, CString};
use std::os::raw::{c_char, c_int};

pub const MC_SPEC_ATOMIC_ADD: u32 = 1;
pub const MC_SPEC_PCAL_INTRO: u32 = 2;
pub const MC_SPEC_RAFT: u32 = 3;
pub const MC_SPEC_SSI: u32 = 4;
pub const MC_SPEC_PCAL: u32 = 5; // a PlusCal algorithm compiled by mc_program_compile
pub const MC_SPEC_PAXOS: u32 = 6; // examples/Paxos/Voting.tla, Paxos.tla under MCVoting / MCPaxos (+ .cfg)
pub const MC_F_GENERIC: u32 = 128;
pub const MC_F_DEADLOCK: u32 = 1;
pub const MC_F_TRACE: u32 = 2;
pub const MC_F_JIT: u32 = 262144; // MC_SPEC_PCAL: the compiled program as generated code, built for the device when the engine is created
pub const MC_MAX_LEVELS: usize = 4096;

#[repr(C)]
pub struct mc_spec_desc { pub spec_id: u32, pub nparams: u32, pub params: [i64; 16] }
#[repr(C)]
pub struct mc_config {
    pub device: i32, pub flags: u32, pub table_capacity: u64, pub arena_capacity: u64, pub chunk_states: u64,
    pub max_levels: u64, pub max_distinct: u64, pub shard_rank: u32, pub shard_count: u32,
}
#[repr(C)]
pub struct mc_result {
    pub distinct: u64, pub generated: u64, pub queue_left: u64, pub depth: u32, pub verdict: i32,
    pub violated_invariant: i32, pub trace_len: u32, pub levels: u32, pub host_evaluated: u32, pub unchecked_properties: u32, pub seconds: f64,
    pub level_distinct: [u64; MC_MAX_LEVELS],
}
#[repr(C)]
pub struct mc_engine { _private: [u8; 0] }
#[repr(C)]
pub struct mc_program { _private: [u8; 0] }

extern "C" {
    pub fn mc_engine_create(spec: *const mc_spec_desc, cfg: *const mc_config, out: *mut *mut mc_engine) -> c_int;
    pub fn mc_engine_run(e: *mut mc_engine, out: *mut mc_result) -> c_int;
    pub fn mc_engine_step(e: *mut mc_engine, levels: u32, out: *mut mc_result) -> c_int;
    pub fn mc_engine_request_stop(e: *mut mc_engine) -> c_int;
    pub fn mc_engine_trace(e: *mut mc_engine, states: *mut u8, actions: *mut i32, n_inout: *mut usize) -> c_int;
    pub fn mc_engine_read_states(e: *mut mc_engine, first: u64, count: u64, out: *mut u8) -> c_int;
    // TLC's checkpoint / -recover (testout1:10): write / reload the states found so far; the next run continues
    pub fn mc_engine_checkpoint(e: *mut mc_engine, path: *const c_char) -> c_int;
    pub fn mc_engine_restore(e: *mut mc_engine, path: *const c_char) -> c_int;
    pub fn mc_engine_destroy(e: *mut mc_engine);
    // one checkpoint file per rank of a sharded run (include/tlamc.h: mc_shard_checkpoint / mc_shard_restore)
    pub fn mc_shard_checkpoint(e: *mut mc_engine, path: *const c_char) -> c_int;
    pub fn mc_shard_restore(e: *mut mc_engine, path: *const c_char) -> c_int;
    // X.tla + X.cfg -> the descriptor mc_engine_create takes (one engine per rank in sharded mode)
    pub fn mc_resolve_files(tla: *const c_char, cfg_path: *const c_char, flags: u32, out: *mut mc_spec_desc,
                            prog_out: *mut *mut mc_program) -> c_int;
    pub fn mc_check_files(tla: *const c_char, cfg_path: *const c_char, cfg: *const mc_config, report: *mut c_char,
                          cap: usize, out: *mut mc_result) -> c_int;
    // PlusCal front-end: `pcal2tla` and the compiler to the GPU interpreter (include/tlamc.h)
    pub fn mc_pcal_translate(tla_text: *const c_char, out: *mut c_char, cap: usize) -> c_int;
    pub fn mc_program_compile(tla_text: *const c_char, cfg_text: *const c_char, out: *mut *mut mc_program) -> c_int;
    pub fn mc_program_spec(p: *const mc_program, out: *mut mc_spec_desc) -> c_int;
    pub fn mc_program_translated(p: *const mc_program) -> *const c_char;
    pub fn mc_program_codegen(p: *const mc_program, buf: *mut c_char, cap: usize) -> std::os::raw::c_long;
    pub fn mc_program_invariant(p: *const mc_program, index: c_int) -> *const c_char;
    pub fn mc_program_free(p: *mut mc_program);
    pub fn mc_state_bytes(spec: *const mc_spec_desc) -> usize;
    pub fn mc_state_format(spec: *const mc_spec_desc, state: *const u8, buf: *mut c_char, cap: usize) -> c_int;
    pub fn mc_action_name(spec: *const mc_spec_desc, action: i32) -> *const c_char;
    pub fn mc_strerror(code: c_int) -> *const c_char;
    pub fn mc_last_error() -> *const c_char;
    pub fn mc_device_count() -> c_int;
}

#[derive(Debug)]
pub struct Outcome { pub distinct: u64, pub generated: u64, pub depth: u32, pub verdict: i32, pub host_evaluated: bool,
                     /// cfg PROPERTIES with a liveness part that were NOT checked (named in the report's "Warning:" line)
                     pub unchecked_properties: u32, pub report: String }

/// `tlc X.tla` (reference Makefile:6-7) from Rust.
pub fn check(tla: &std::path::Path, device: i32) -> Result<Outcome, String> {
    let path = CString::new(tla.to_str().ok_or("non-UTF-8 path")?).map_err(|e| e.to_string())?;
    let cfg = mc_config { device, flags: MC_F_DEADLOCK | MC_F_TRACE, table_capacity: 1 << 26, arena_capacity: 1 << 24,
                          chunk_states: 0, max_levels: 0, max_distinct: 0, shard_rank: 0, shard_count: 1 };
    let mut res: Box<mc_result> = unsafe { Box::new(std::mem::zeroed()) };
    let mut buf = vec![0u8; 1 << 20];
    let rc = unsafe { mc_check_files(path.as_ptr(), std::ptr::null(), &cfg, buf.as_mut_ptr() as *mut c_char, buf.len(), &mut *res) };
    if rc != 0 {
        return Err(unsafe { CStr::from_ptr(mc_last_error()) }.to_string_lossy().into_owned());
    }
    let end = buf.iter().position(|&b| b == 0).unwrap_or(0);
    Ok(Outcome { distinct: res.distinct, generated: res.generated, depth: res.depth, verdict: res.verdict, host_evaluated: res.host_evaluated != 0, unchecked_properties: res.unchecked_properties,
                 report: String::from_utf8_lossy(&buf[..end]).into_owned() })
}

#[cfg(test)]
mod tests {
    /// What the README's tutorial promises once the labels A:/B: are removed (README.md:349-352).
    #[test]
    fn committed_pcal_intro_has_no_error() {
        let o = super::check(std::path::Path::new("../../specs/pcal_intro.tla"), 0).unwrap();
        assert_eq!((o.verdict, o.distinct, o.generated, o.depth), (0, 3800, 5850, 5));
    }
}
