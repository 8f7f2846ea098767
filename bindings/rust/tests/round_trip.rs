//! The crate's integration test (reference tests/test_zlib.rs:30-115: compress -> decompress identity with Fixed,
//! Dynamic and Auto on LICENSE and on every file under src/) against this shim.  The inputs are generated here --
//! text of a licence's size and source-like text of the sizes of the crate's files (167 B ... 35 875 B) -- not
//! copied from the crate.
use zlib_amd_shim::huffman::HuffmanTree;
use zlib_amd_shim::lz77::{LZ77Compressor, LZ77Unit};
use zlib_amd_shim::{compress, decompress, Strategy};

fn words(seed: u64, n: usize) -> Vec<u8> {
    let vocab = [
        "fn", "let", "mut", "pub", "self", "usize", "return", "match", "impl", "the", "of", "and", "to", "in",
        "permission", "software", "without", "warranty", "copyright", "notice", "shall", "be", "included",
    ];
    let (mut s, mut out) = (seed | 1, Vec::with_capacity(n + 16));
    while out.len() < n {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        out.extend_from_slice(vocab[(s % vocab.len() as u64) as usize].as_bytes());
        out.push(if s % 11 == 0 { b'\n' } else { b' ' });
    }
    out.truncate(n);
    out
}

fn inputs() -> Vec<Vec<u8>> {
    // LICENSE (1081 B) and the sizes of the files under src/
    [1081usize, 167, 327, 4705, 6141, 10324, 13939, 30522, 35875].iter().enumerate().map(|(i, &n)| words(i as u64 + 7, n)).collect()
}

fn round_trip(strategy: &Strategy) {
    for bytes in inputs() {
        let compressed = compress(&bytes, strategy);
        let decompressed = decompress(&compressed).expect("Correct decompression");
        assert_eq!(bytes, decompressed);
        // the crate's own decoder reads the same stream
        assert_eq!(zlib::decompress(&compressed).expect("Correct decompression"), bytes);
    }
}

#[test]
fn test_fixed() {
    round_trip(&Strategy::Fixed);
}

#[test]
fn test_dynamic() {
    round_trip(&Strategy::Dynamic);
}

#[test]
fn test_auto() {
    round_trip(&Strategy::Auto);
}

#[test]
fn test_units_and_trees_equal_the_crates_own() {
    let data = words(99, 20_000);
    let mut own = zlib::lz77::LZ77Compressor::with_window_size(1 << 15);
    own.min_match_length = 3;
    own.max_match_length = 258;
    let mut gpu = LZ77Compressor::with_window_size(1 << 15);
    gpu.min_match_length = 3;
    gpu.max_match_length = 258;
    let units: Vec<LZ77Unit> = gpu.compress(&data);
    assert_eq!(units, own.compress(&data));
    assert_eq!(LZ77Compressor::with_window_size((1 << 15) + 1).window_size, 1 << 15);
    // Code tables.  The crate's heap leaves the order of two internal nodes of equal weight open (FreqNode::cmp,
    // src/huffman.rs:103-116, over a HashMap's iteration order), so its code LENGTHS may differ between two runs in
    // such a tie; whatever the order, both trees are optimal prefix codes over the same symbols.  Checked here:
    // the same symbols, the same total cost (sum of count x length), a full Kraft sum on both sides -- and the same
    // maps outright when the lengths agree, i.e. whenever no such tie was met (canonical codes follow from the
    // lengths alone).
    let (mut lt, mut dt) = HuffmanTree::from_lz77(&units, &gpu);
    lt.assign();
    dt.assign();
    let (mut lo, mut d_o) = zlib::huffman::HuffmanTree::from_lz77(&units, &own);
    lo.assign();
    d_o.assign();
    let (lo, d_o) = (lo.to_canonical(), d_o.to_canonical());
    let (lt, dt) = (lt.to_canonical(), dt.to_canonical());
    let (mut lit_freq, mut dist_freq) = (std::collections::HashMap::new(), std::collections::HashMap::new());
    for u in &units {
        match u {
            LZ77Unit::Literal(b) => *lit_freq.entry(char::from(*b)).or_insert(0usize) += 1,
            LZ77Unit::Marker(len, dist) => {
                let lc = char::from_u32(zlib::huffman::get_length_code(*len) as u32).unwrap();
                let dc = char::from_u32(zlib::huffman::get_distance_code(*dist) as u32).unwrap();
                *lit_freq.entry(lc).or_insert(0usize) += 1;
                *dist_freq.entry(dc).or_insert(0usize) += 1;
            }
        }
    }
    lit_freq.entry(char::from_u32(256).unwrap()).or_insert(1); // the end-of-block symbol, src/huffman.rs:264
    type Enc = std::collections::HashMap<char, (usize, usize)>;
    let same_code = |mine: &Enc, theirs: &Enc, freq: &std::collections::HashMap<char, usize>| {
        let mut a: Vec<&char> = mine.keys().collect();
        let mut b: Vec<&char> = theirs.keys().collect();
        a.sort();
        b.sort();
        assert_eq!(a, b, "the two trees code different symbols");
        let cost = |e: &Enc| freq.iter().map(|(s, f)| f * e[s].1).sum::<usize>();
        assert_eq!(cost(mine), cost(theirs), "the two trees differ in total cost");
        let kraft = |e: &Enc| e.values().map(|&(_, l)| 1u64 << (32 - l)).sum::<u64>();
        if mine.len() > 1 {
            assert_eq!(kraft(mine), 1u64 << 32);
            assert_eq!(kraft(theirs), 1u64 << 32);
        }
        if mine.iter().all(|(s, &(_, l))| theirs[s].1 == l) {
            assert_eq!(mine, theirs, "equal lengths must give equal canonical codes");
        }
    };
    same_code(lt.encodings().unwrap(), lo.encodings().unwrap(), &lit_freq);
    same_code(dt.encodings().unwrap(), d_o.encodings().unwrap(), &dist_freq);
}

#[test]
fn test_stream_front_end_gives_the_bytes_of_compress() {
    // new items of the shim: the same stream from a Read into a Write, and between two paths
    let data = words(7, 3 << 20);
    let want = compress(&data, &Strategy::Dynamic);
    let mut got = Vec::new();
    let (n_in, n_out) = zlib_amd_shim::compress::compress_stream(&mut &data[..], &mut got, &Strategy::Dynamic).unwrap();
    assert_eq!((n_in as usize, n_out as usize), (data.len(), got.len()));
    assert_eq!(got, want);
    let dir = std::env::temp_dir();
    let (src, dst) = (dir.join("zlib_amd_stream_in.bin"), dir.join("zlib_amd_stream_out.z"));
    std::fs::write(&src, &data).unwrap();
    zlib_amd_shim::compress::compress_file(&src, &dst, &Strategy::Dynamic).unwrap();
    assert_eq!(std::fs::read(&dst).unwrap(), want);
}
