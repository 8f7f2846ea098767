//! The compress-side public items of the `zlib` crate with the same names, argument meaning and error
//! behaviour (reference: src/compress.rs:28-34,50-51; src/lz77.rs:23-120,143; src/adler.rs:5;
//! src/decompress.rs:34), served by libzlib_amd.so.  A maintainer replaces the bodies of the crate's own
//! functions with these; everything that is not on the path (BitReader, BitWriter, the pointer-tree
//! HuffmanTree API) stays the crate's Rust.
//!
//! NOT compiled in the image this repository was developed in (no rustc/cargo there).
pub mod ffi;
use ffi::check;

#[derive(Debug)]
pub enum Strategy {
    Auto,
    Dynamic,
    Fixed,
    Raw,
}

fn code(s: &Strategy) -> i32 {
    match s {
        Strategy::Auto => 0,
        Strategy::Dynamic => 1,
        Strategy::Fixed => 2,
        Strategy::Raw => 3,
    }
}

unsafe fn take(out: *mut u8, n: usize) -> Vec<u8> {
    let v = std::slice::from_raw_parts(out, n).to_vec();
    ffi::zg_free(out.cast());
    v
}

/// zlib::compress (src/compress.rs:50-51)
#[must_use]
pub fn compress(data: &[u8], strategy: &Strategy) -> Vec<u8> {
    let (mut out, mut n) = (std::ptr::null_mut::<u8>(), 0usize);
    check(unsafe {
        ffi::zg_compress(std::ptr::null_mut(), data.as_ptr(), data.len(), code(strategy), std::ptr::null(), &mut out, &mut n)
    });
    unsafe { take(out, n) }
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

/// zlib::decompress (src/decompress.rs:34): header and trailer faults are Err(String); the device decoder
/// reports corrupt bodies the same way instead of panicking.
pub fn decompress(input: &[u8]) -> Result<Vec<u8>, String> {
    let (mut out, mut n) = (std::ptr::null_mut::<u8>(), 0usize);
    let rc = unsafe { ffi::zg_decompress(std::ptr::null_mut(), input.as_ptr(), input.len(), &mut out, &mut n) };
    if rc == ffi::ZG_ERR_DATA {
        return Err(unsafe { std::ffi::CStr::from_ptr(ffi::zg_last_error(std::ptr::null_mut())) }
            .to_string_lossy()
            .into_owned());
    }
    check(rc);
    Ok(unsafe { take(out, n) })
}

pub mod adler {
    /// zlib::adler::adler32 (src/adler.rs:5-13)
    pub fn adler32(data: &[u8]) -> u32 {
        let mut v = 0u32;
        crate::ffi::check(unsafe { crate::ffi::zg_adler32(std::ptr::null_mut(), data.as_ptr(), data.len(), &mut v) });
        v
    }
}

pub mod lz77 {
    use crate::ffi;

    /// src/lz77.rs:23-27
    #[derive(Debug, PartialEq, Eq, PartialOrd, Ord)]
    pub enum LZ77Unit {
        Literal(u8),
        Marker(usize, usize),
    }

    /// src/lz77.rs:31-38; the constructors keep the crate's defaults (144 / 5 / 100, src/lz77.rs:64-75)
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
        pub fn new() -> Self {
            Self { window_size: 144, min_match_length: 5, max_match_length: 100 }
        }
        pub fn with_window_size(window_size: usize) -> Self {
            Self { window_size, ..Self::new() }
        }
        pub fn set_window_size(&mut self, window_size: usize) -> &mut Self {
            self.window_size = window_size;
            self
        }
        /// src/lz77.rs:143-207.  The device path needs 3 <= min_match_length and max_match_length <= 258.
        pub fn compress(&self, data: &[u8]) -> Vec<LZ77Unit> {
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

    /// The code lengths and LSB-first codes HuffmanTree::from_lz77 + assign + to_canonical arrive at
    /// (src/huffman.rs:238-288,619-645,714-734); the crate builds its two pointer trees from the lengths with its
    /// own from_bitlen_list (src/huffman.rs:311-337).
    pub struct CodeTables {
        pub ll_len: [u8; 286],
        pub ll_code: [u16; 286],
        pub d_len: [u8; 30],
        pub d_code: [u16; 30],
    }

    pub fn tables_from_lz77(units: &[lz77::LZ77Unit]) -> CodeTables {
        let toks = lz77::pack(units);
        let mut t = CodeTables { ll_len: [0; 286], ll_code: [0; 286], d_len: [0; 30], d_code: [0; 30] };
        ffi::check(unsafe {
            ffi::zg_huffman_from_tokens(std::ptr::null_mut(), toks.as_ptr(), toks.len(), t.ll_len.as_mut_ptr(),
                                        t.ll_code.as_mut_ptr(), t.d_len.as_mut_ptr(), t.d_code.as_mut_ptr())
        });
        t
    }
}

#[cfg(test)]
mod tests {
    // the reference's own round-trip checks (tests/test_zlib.rs:30-115) in miniature
    use super::*;
    #[test]
    fn round_trip() {
        let data: Vec<u8> = (0..200_000u32).map(|i| (i % 251) as u8 ^ (i / 7) as u8).collect();
        for s in [Strategy::Auto, Strategy::Dynamic, Strategy::Fixed, Strategy::Raw] {
            assert_eq!(decompress(&compress(&data, &s)).unwrap(), data);
        }
        assert_eq!(adler::adler32(b"compressed"), 0x173F0436);
    }
}
