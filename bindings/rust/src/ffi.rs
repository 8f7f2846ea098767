//! Declarations of include/zlib_amd.h (the C ABI): the entry points this crate calls.
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct ZgOpts {
    pub chunk_bytes: usize,
    pub batch_bytes: usize,
}

pub const ZG_ERR_DATA: c_int = -8;
pub const ZG_ERR_IO: c_int = -9;

pub type ZgReadFn = unsafe extern "C" fn(user: *mut c_void, buf: *mut u8, cap: usize) -> usize;
pub type ZgWriteFn = unsafe extern "C" fn(user: *mut c_void, buf: *const u8, n: usize) -> c_int;

extern "C" {
    pub fn zg_strerror(code: c_int) -> *const c_char;
    pub fn zg_last_error(ctx: *mut c_void) -> *const c_char;
    pub fn zg_free(p: *mut c_void);
    pub fn zg_compress(ctx: *mut c_void, input: *const u8, n: usize, strategy: c_int, opts: *const ZgOpts,
                       out: *mut *mut u8, out_n: *mut usize) -> c_int;
    pub fn zg_compress_multi(devices: *const c_int, n_devices: c_int, input: *const u8, n: usize, strategy: c_int,
                             opts: *const ZgOpts, out: *mut *mut u8, out_n: *mut usize) -> c_int;
    pub fn zg_compress_stream(ctx: *mut c_void, read_fn: ZgReadFn, read_user: *mut c_void, write_fn: ZgWriteFn,
                              write_user: *mut c_void, strategy: c_int, opts: *const ZgOpts, size_hint: u64,
                              in_total: *mut u64, out_total: *mut u64) -> c_int;
    pub fn zg_compress_file(ctx: *mut c_void, in_path: *const c_char, out_path: *const c_char, strategy: c_int,
                            opts: *const ZgOpts, in_bytes: *mut u64, out_bytes: *mut u64) -> c_int;
    pub fn zg_compress_batch(ctx: *mut c_void, input: *const u8, offsets: *const usize, n_records: usize,
                             strategy: c_int, out: *mut *mut u8, out_offsets: *mut *mut usize) -> c_int;
    pub fn zg_decompress(ctx: *mut c_void, input: *const u8, n: usize, out: *mut *mut u8, out_n: *mut usize) -> c_int;
    pub fn zg_adler32(ctx: *mut c_void, input: *const u8, n: usize, out: *mut u32) -> c_int;
    pub fn zg_lz77(ctx: *mut c_void, input: *const u8, n: usize, start: usize, window: u32, min_len: u32,
                   max_len: u32, tokens: *mut *mut u32, n_tokens: *mut usize) -> c_int;
    pub fn zg_huffman_from_tokens(ctx: *mut c_void, tokens: *const u32, n: usize, ll_len: *mut u8,
                                  ll_code: *mut u16, d_len: *mut u8, d_code: *mut u16) -> c_int;
}

/// compress() and LZ77Compressor::compress() of the crate signal failure by panic.
pub fn check(rc: c_int) {
    if rc != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(zg_strerror(rc)) }.to_string_lossy().into_owned();
        panic!("zlib_amd: {msg} ({rc})");
    }
}
