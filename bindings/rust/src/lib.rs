//! The `zlib` crate's public surface under its own module names (reference: src/lib.rs:1-10), with the compress
//! path served by libzlib_amd.so:
//!
//!   accelerated here      compress (src/compress.rs:50-51), decompress (src/decompress.rs:34),
//!                         adler::adler32 (src/adler.rs:5), lz77::LZ77Compressor::compress (src/lz77.rs:143),
//!                         huffman::HuffmanTree::from_lz77 (src/huffman.rs:238)
//!   the crate's own code  bitreader, bitwriter, every other item of huffman (the pointer tree with insert /
//!                         decode / decode_trees / assign / encode / to_canonical, the tables and constants),
//!                         lz77::LZ77Unit, Strategy
//!
//! so that `use zlib_amd_shim::{compress, Strategy}`, `use zlib_amd_shim::huffman::HuffmanTree`,
//! `use zlib_amd_shim::bitwriter::BitWriter` ... resolve exactly as they do against the crate.
//!
//! NOT compiled in the image this repository was developed in (it has neither rustc nor cargo): written as
//! source for a machine that has both; `cargo test` there runs tests/round_trip.rs, the crate's own integration
//! test (tests/test_zlib.rs:30-115) on generated inputs.
pub mod ffi;
use ffi::check;

pub use zlib::bitreader;
pub use zlib::bitwriter;

pub mod compress {
    use crate::ffi::{self, check};
    pub use zlib::Strategy;

    pub(crate) fn code(s: &Strategy) -> i32 {
        match s {
            Strategy::Auto => 0,
            Strategy::Dynamic => 1,
            Strategy::Fixed => 2,
            Strategy::Raw => 3,
        }
    }

    pub(crate) unsafe fn take(out: *mut u8, n: usize) -> Vec<u8> {
        let v = std::slice::from_raw_parts(out, n).to_vec();
        ffi::zg_free(out.cast());
        v
    }

    /// zlib::compress (src/compress.rs:50-51).  Inputs of more than one chunk (1 MiB) come back as one block
    /// per chunk; every inflater, the crate's decompress included, reads them as one stream.
    #[must_use]
    pub fn compress(data: &[u8], strategy: &Strategy) -> Vec<u8> {
        let (mut out, mut n) = (std::ptr::null_mut::<u8>(), 0usize);
        check(unsafe {
            ffi::zg_compress(std::ptr::null_mut(), data.as_ptr(), data.len(), code(strategy), std::ptr::null(), &mut out, &mut n)
        });
        unsafe { take(out, n) }
    }

    /// New item: the same stream from several GPUs of this process (HIP device indices).
    #[must_use]
    pub fn compress_on(devices: &[i32], data: &[u8], strategy: &Strategy) -> Vec<u8> {
        let (mut out, mut n) = (std::ptr::null_mut::<u8>(), 0usize);
        check(unsafe {
            ffi::zg_compress_multi(devices.as_ptr(), devices.len() as i32, data.as_ptr(), data.len(), code(strategy),
                                   std::ptr::null(), &mut out, &mut n)
        });
        unsafe { take(out, n) }
    }

    /// New item: compress from any `Read` into any `Write` without holding the whole input in memory; the bytes
    /// written are those `compress` returns for the whole input.  Returns (input bytes, stream bytes).
    /// The library's default context is locked for the whole call and `reader` / `writer` run on threads of the
    /// library: a reader or writer that itself calls into this crate (`compress`, `adler32`, ...) would wait for
    /// that lock for ever.
    pub fn compress_stream<R: std::io::Read + Send, W: std::io::Write + Send>(
        reader: &mut R, writer: &mut W, strategy: &Strategy,
    ) -> std::io::Result<(u64, u64)> {
        use std::os::raw::{c_int, c_void};
        // One end per callback: the library calls `pull` from one of its threads and `push` from another, each
        // from that thread only and never after zg_compress_stream has returned, so the two never share state.
        struct End<'a, T> {
            io: &'a mut T,
            err: Option<std::io::Error>,
        }
        unsafe extern "C" fn pull<R: std::io::Read>(user: *mut c_void, buf: *mut u8, cap: usize) -> usize {
            let end = &mut *(user as *mut End<R>);
            loop {
                match end.io.read(std::slice::from_raw_parts_mut(buf, cap)) {
                    Ok(n) => return n,
                    Err(e) if e.kind() == std::io::ErrorKind::Interrupted => continue,
                    Err(e) => {
                        end.err = Some(e);
                        return usize::MAX;
                    }
                }
            }
        }
        unsafe extern "C" fn push<W: std::io::Write>(user: *mut c_void, buf: *const u8, n: usize) -> c_int {
            let end = &mut *(user as *mut End<W>);
            match end.io.write_all(std::slice::from_raw_parts(buf, n)) {
                Ok(()) => 0,
                Err(e) => {
                    end.err = Some(e);
                    -1
                }
            }
        }
        let mut src = End { io: reader, err: None };
        let mut dst = End { io: writer, err: None };
        let (mut n_in, mut n_out) = (0u64, 0u64);
        let rc = unsafe {
            ffi::zg_compress_stream(std::ptr::null_mut(), pull::<R>, &mut src as *mut End<R> as *mut c_void, push::<W>,
                                    &mut dst as *mut End<W> as *mut c_void, code(strategy), std::ptr::null(), 0,
                                    &mut n_in, &mut n_out)
        };
        if let Some(e) = src.err.take().or(dst.err.take()) {
            return Err(e);
        }
        check(rc);
        Ok((n_in, n_out))
    }

    /// New item: `compress_stream` between two paths, read and written by the library itself.
    pub fn compress_file(input: &std::path::Path, output: &std::path::Path, strategy: &Strategy) -> std::io::Result<(u64, u64)> {
        use std::os::unix::ffi::OsStrExt;
        let a = std::ffi::CString::new(input.as_os_str().as_bytes())?;
        let b = std::ffi::CString::new(output.as_os_str().as_bytes())?;
        let (mut n_in, mut n_out) = (0u64, 0u64);
        let rc = unsafe {
            ffi::zg_compress_file(std::ptr::null_mut(), a.as_ptr(), b.as_ptr(), code(strategy), std::ptr::null(), &mut n_in, &mut n_out)
        };
        if rc == ffi::ZG_ERR_IO {
            let msg = unsafe { std::ffi::CStr::from_ptr(ffi::zg_last_error(std::ptr::null_mut())) }.to_string_lossy().into_owned();
            return Err(std::io::Error::new(std::io::ErrorKind::Other, msg));
        }
        check(rc);
        Ok((n_in, n_out))
    }

    /// New item: every record compressed as a zlib stream of its own, all in flight together.
    pub fn compress_many(records: &[&[u8]], strategy: &Strategy) -> Vec<Vec<u8>> {
        if !matches!(strategy, Strategy::Dynamic | Strategy::Fixed) {
            return records.iter().map(|r| compress(r, strategy)).collect();
        }
        let mut blob = Vec::new();
        let mut offs = vec![0usize];
        for r in records {
            blob.extend_from_slice(r);
            offs.push(blob.len());
        }
        let (mut out, mut oo) = (std::ptr::null_mut::<u8>(), std::ptr::null_mut::<usize>());
        check(unsafe {
            ffi::zg_compress_batch(std::ptr::null_mut(), blob.as_ptr(), offs.as_ptr(), records.len(), code(strategy), &mut out, &mut oo)
        });
        let o = unsafe { std::slice::from_raw_parts(oo, records.len() + 1) };
        let v = (0..records.len())
            .map(|i| unsafe { std::slice::from_raw_parts(out.add(o[i]), o[i + 1] - o[i]) }.to_vec())
            .collect();
        unsafe {
            ffi::zg_free(out.cast());
            ffi::zg_free(oo.cast());
        }
        v
    }
}

pub mod decompress {
    use crate::ffi::{self, check};

    /// zlib::decompress (src/decompress.rs:34): header and trailer faults are Err(String) with the crate's own
    /// messages; the device decoder reports corrupt bodies the same way instead of panicking.
    pub fn decompress(input: &[u8]) -> Result<Vec<u8>, String> {
        let (mut out, mut n) = (std::ptr::null_mut::<u8>(), 0usize);
        let rc = unsafe { ffi::zg_decompress(std::ptr::null_mut(), input.as_ptr(), input.len(), &mut out, &mut n) };
        if rc == ffi::ZG_ERR_DATA {
            return Err(unsafe { std::ffi::CStr::from_ptr(ffi::zg_last_error(std::ptr::null_mut())) }
                .to_string_lossy()
                .into_owned());
        }
        check(rc);
        Ok(unsafe { crate::compress::take(out, n) })
    }
}

// src/lib.rs:9-10
pub use compress::*;
pub use decompress::*;

pub mod adler {
    /// zlib::adler::adler32 (src/adler.rs:5-13)
    pub fn adler32(data: &[u8]) -> u32 {
        let mut v = 0u32;
        crate::check(unsafe { crate::ffi::zg_adler32(std::ptr::null_mut(), data.as_ptr(), data.len(), &mut v) });
        v
    }
}

pub mod lz77 {
    use crate::ffi;
    pub use zlib::lz77::LZ77Unit;

    const MAX_WINDOW_SIZE: usize = 1 << 15; // src/lz77.rs:20

    /// src/lz77.rs:30-38: the same public fields, defaults (144 / 5 / 100, src/lz77.rs:18-21,64-75) and clamping.
    #[derive(Debug)]
    pub struct LZ77Compressor {
        pub window_size: usize,
        pub min_match_length: usize,
        pub max_match_length: usize,
    }

    impl Default for LZ77Compressor {
        fn default() -> Self {
            Self::new()
        }
    }

    impl LZ77Compressor {
        #[must_use]
        pub fn new() -> Self {
            Self { window_size: 144, min_match_length: 5, max_match_length: 100 }
        }
        /// src/lz77.rs:90-96: the window is clamped to 32 KiB
        #[must_use]
        pub fn with_window_size(window_size: usize) -> Self {
            Self { window_size: window_size.min(MAX_WINDOW_SIZE), ..Self::new() }
        }
        /// src/lz77.rs:117-120
        pub fn set_window_size(&mut self, newsize: usize) -> &mut Self {
            self.window_size = newsize.min(MAX_WINDOW_SIZE);
            self
        }
        /// src/lz77.rs:143-207.  The device path needs 3 <= min_match_length and max_match_length <= 258;
        /// other parameters fall through to the crate's own loop.
        #[must_use]
        pub fn compress(&self, data: &[u8]) -> Vec<LZ77Unit> {
            if self.min_match_length < 3 || self.max_match_length > 258 {
                let mut own = zlib::lz77::LZ77Compressor::with_window_size(self.window_size);
                own.min_match_length = self.min_match_length;
                own.max_match_length = self.max_match_length;
                return own.compress(data);
            }
            let (mut t, mut n) = (std::ptr::null_mut::<u32>(), 0usize);
            ffi::check(unsafe {
                ffi::zg_lz77(std::ptr::null_mut(), data.as_ptr(), data.len(), 0, self.window_size as u32,
                             self.min_match_length as u32, self.max_match_length as u32, &mut t, &mut n)
            });
            let units = unsafe { std::slice::from_raw_parts(t, n) }
                .iter()
                .map(|&u| {
                    if u & 0x8000_0000 != 0 {
                        LZ77Unit::Marker(((u >> 16) & 0x1FF) as usize, (u & 0xFFFF) as usize)
                    } else {
                        LZ77Unit::Literal(u as u8)
                    }
                })
                .collect();
            unsafe { ffi::zg_free(t.cast()) };
            units
        }
    }

    /// Packs units the way zg_huffman_from_tokens reads them (marker: bit 31 | length << 16 | distance).
    pub fn pack(units: &[LZ77Unit]) -> Vec<u32> {
        units
            .iter()
            .map(|u| match *u {
                LZ77Unit::Literal(b) => b as u32,
                LZ77Unit::Marker(len, dist) => 0x8000_0000 | ((len as u32) << 16) | dist as u32,
            })
            .collect()
    }
}

pub mod huffman {
    use crate::{ffi, lz77};
    use std::collections::HashMap;
    use std::ops::{Deref, DerefMut};
    use zlib::bitreader::BitReader;

    // the tables, constants and helper functions are the crate's own (src/huffman.rs:26-49,785-871)
    pub use zlib::huffman::{
        distance_tree_alphabet, get_distance_code, get_length_code, literal_length_tree_alphabet,
        CODE_LENGTH_CODES_ORDER, DISTANCE_BASE, DISTANCE_EXTRA_BITS, LENGTH_BASE, LENGTH_EXTRA_BITS,
        ZLIB_MAX_STRING_LENGTH, ZLIB_MIN_STRING_LENGTH, ZLIB_WINDOW_SIZE,
    };

    /// The crate's pointer tree (src/huffman.rs:66-761) behind a thin wrapper: every method of the crate's type is
    /// reachable through Deref (insert, decode, assign, encode, encodings, to_canonical, n_codes, max_code_len),
    /// the constructors are forwarded, and from_lz77 takes its histograms and code lengths from the GPU.
    #[derive(Debug)]
    pub struct HuffmanTree(pub zlib::huffman::HuffmanTree);

    impl Deref for HuffmanTree {
        type Target = zlib::huffman::HuffmanTree;
        fn deref(&self) -> &Self::Target {
            &self.0
        }
    }
    impl DerefMut for HuffmanTree {
        fn deref_mut(&mut self) -> &mut Self::Target {
            &mut self.0
        }
    }
    impl Default for HuffmanTree {
        fn default() -> Self {
            Self::new()
        }
    }

    impl HuffmanTree {
        #[must_use]
        pub fn new() -> Self {
            Self(zlib::huffman::HuffmanTree::new())
        }
        #[must_use]
        pub fn from_data(data: &[u8]) -> Self {
            Self(zlib::huffman::HuffmanTree::from_data(data))
        }
        #[must_use]
        pub fn from_freq(frequencies: HashMap<char, usize>) -> Self {
            Self(zlib::huffman::HuffmanTree::from_freq(frequencies))
        }
        #[must_use]
        pub fn from_bitlen_list(bitlen: &[usize], alphabet: &[char]) -> Self {
            Self(zlib::huffman::HuffmanTree::from_bitlen_list(bitlen, alphabet))
        }
        #[must_use]
        pub fn get_zlib_fixed() -> (Self, Self) {
            let (a, b) = zlib::huffman::HuffmanTree::get_zlib_fixed();
            (Self(a), Self(b))
        }
        pub fn decode_trees(reader: &mut BitReader) -> (Self, Self) {
            let (a, b) = zlib::huffman::HuffmanTree::decode_trees(reader);
            (Self(a), Self(b))
        }
        #[must_use]
        pub fn to_canonical(&self) -> Self {
            Self(self.0.to_canonical())
        }

        /// HuffmanTree::from_lz77 (src/huffman.rs:238-288).  The histograms and the code lengths come from the
        /// GPU; the two trees are then built from those lengths with the crate's own from_bitlen_list
        /// (src/huffman.rs:311-337).  They are canonical from the start, so `assign` + `to_canonical`, which
        /// compress_dynamic applies next (src/compress.rs:296-314), leave exactly the codes the crate arrives at
        /// (the code LENGTHS are the crate's wherever its heap order is defined: equal-weight internal nodes, which
        /// the crate takes in the order of a randomly seeded HashMap, are taken older first; the total cost is the same).
        /// Panics like check_lz77 (src/huffman.rs:817-835) on other than the zlib parameters.
        pub fn from_lz77(data: &[lz77::LZ77Unit], lz77: &lz77::LZ77Compressor) -> (Self, Self) {
            assert_eq!(lz77.window_size, ZLIB_WINDOW_SIZE, "Incompatible LZ77 window size");
            assert_eq!(lz77.min_match_length, ZLIB_MIN_STRING_LENGTH, "Incompatible LZ77 min string length");
            assert_eq!(lz77.max_match_length, ZLIB_MAX_STRING_LENGTH, "Incompatible LZ77 max string length");
            let toks = lz77::pack(data);
            let (mut ll_len, mut ll_code) = ([0u8; 286], [0u16; 286]);
            let (mut d_len, mut d_code) = ([0u8; 30], [0u16; 30]);
            ffi::check(unsafe {
                ffi::zg_huffman_from_tokens(std::ptr::null_mut(), toks.as_ptr(), toks.len(), ll_len.as_mut_ptr(),
                                            ll_code.as_mut_ptr(), d_len.as_mut_ptr(), d_code.as_mut_ptr())
            });
            let ll: Vec<usize> = ll_len.iter().map(|&l| l as usize).collect();
            let d: Vec<usize> = d_len.iter().map(|&l| l as usize).collect();
            (
                Self::from_bitlen_list(&ll, &literal_length_tree_alphabet()),
                Self::from_bitlen_list(&d, &distance_tree_alphabet()),
            )
        }
    }
}
