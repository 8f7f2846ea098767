"""Huffman codes outside the reference's domain, on the GPU.  The crate panics when a literal/length or distance
code would be longer than 15 bits (src/compress.rs:302-310), when the code-length code would be longer than 7
(:421-426), when a single distance symbol is used and on empty input; the library limits the lengths and gives a
lone symbol one bit.  The wave-wide block builder that k_huffman runs (its length limiting included) must give
what the oracle's build mode gives: tables through zg_huffman_from_tokens, whole streams through zg_compress and
zg_compress_batch.  Every case states, with a plain heap construction, that the unlimited code really is too
deep -- the test would otherwise pass without reaching the code it is about."""
import heapq
import random
import zlib as sysz

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def za():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    torch.zeros(1, device="cuda")
    import zlib_amd
    zlib_amd.default_context()
    return zlib_amd


@pytest.fixture(scope="module")
def z():
    from oracle import zref
    zref.lib()
    return zref


def _depth(freqs):
    """Longest code of an unlimited Huffman code over the non-zero counts (any tie-break gives an optimal tree; the
    depth may differ by tie-break only when it is far from the limits used below)."""
    h = [(f, i, 0) for i, f in enumerate(freqs) if f]
    if len(h) < 2:
        return len(h)
    heapq.heapify(h)
    k = len(freqs)
    while len(h) > 1:
        a, b = heapq.heappop(h), heapq.heappop(h)
        heapq.heappush(h, (a[0] + b[0], k, max(a[2], b[2]) + 1))
        k += 1
    return h[0][2]


def _hist(z, toks):
    ll, d = [0] * 286, [0] * 30
    for t in toks:
        if t & z.TOK_MARKER:
            ll[z.get_length_code((t >> 16) & 0x1FF)] += 1
            d[z.get_distance_code(t & 0xFFFF)] += 1
        else:
            ll[t & 0xFF] += 1
    ll[256] = max(1, ll[256])
    return ll, d


def _cl_depth(z, toks):
    """Depth of the unlimited code over the run-length symbols of the block's header (get_code_tree)."""
    ll_len, _, d_len, _ = z.chunk_tables(toks)
    nl = max([257] + [i for i in range(286) if ll_len[i]]) + 1
    used = [i for i in range(30) if d_len[i]]
    nd = (max(used) if used else 1) + 1
    h = [0] * 19
    for s, _ in z.zlib_rle(ll_len[:nl], d_len[:nd]):
        h[s] += 1
    return _depth(h)


FIB = [1, 1]
while len(FIB) < 40:
    FIB.append(FIB[-1] + FIB[-2])

DISTS = [5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145]


def _stress_tokens(z, seed):
    rng = random.Random(seed)
    toks = []
    for s in range(256):
        toks += [s] * rng.choice([0, 1, 1, 2, 4, 9, 30, 200, 3000])
    for _ in range(rng.randrange(0, 300)):
        toks.append(z.marker(rng.randrange(3, 258), rng.randrange(5, 32769)))
    return toks


def _check_tables(za, z, toks, what):
    lt, dt = za.HuffmanTree.from_lz77(toks, za.zlib_compressor())
    ll_len, ll_code, d_len, d_code = z.chunk_tables(toks)
    assert lt.encodings() == {s: (ll_code[s], ll_len[s]) for s in range(286) if ll_len[s]}, what
    assert dt.encodings() == {s: (d_code[s], d_len[s]) for s in range(30) if d_len[s]}, what
    return lt, dt


def test_tables_beyond_the_reference_domain(za, z):
    # lone distance symbol, no unit at all, markers only (no literal but the end-of-block symbol)
    lt, dt = _check_tables(za, z, [97] * 10 + [z.marker(3, 7)], "lone distance")
    assert dt.encodings() == {z.get_distance_code(7): (0, 1)}
    lt, dt = _check_tables(za, z, [], "empty")
    assert lt.encodings() == {256: (0, 1)} and dt.encodings() == {}
    _check_tables(za, z, [z.marker(3, 9)] * 5, "markers only")
    # Fibonacci counts (the end-of-block symbol supplies the second 1): literal/length code deeper than 15
    toks = []
    for s, c in enumerate(FIB[1:27]):
        toks += [s] * c
    assert _depth(_hist(z, toks)[0]) > 20
    lt, _ = _check_tables(za, z, toks, "Fibonacci literals")
    assert lt.max_code_len() == 15
    # ... and a distance code deeper than 15, beside a shallow literal/length code
    toks = [1, 2, 3]
    for i, dd in enumerate(DISTS):
        toks += [z.marker(3 + i, dd)] * FIB[i]
    assert _depth(_hist(z, toks)[1]) > 15
    _, dt = _check_tables(za, z, toks, "Fibonacci distances")
    assert dt.max_code_len() == 15
    # both at once, all 286 + 30 symbols in use
    toks = []
    for s in range(256):
        toks += [s] * FIB[1 + s % 24]
    for i in range(30):
        toks += [z.marker(z.table("length_base")[i % 29], z.table("distance_base")[i])] * (FIB[i + 1] if i < 20 else 20000)
    toks += [z.marker(3, 32768)] * 7
    ll, d = _hist(z, toks)
    assert _depth(ll) > 15 and _depth(d) > 15
    _check_tables(za, z, toks, "both alphabets")
    # histograms whose header needs a code-length code deeper than 7 (tables are not affected, the streams below are)
    deep = 0
    for seed in range(30):
        toks = _stress_tokens(z, seed)
        deep += _cl_depth(z, toks) > 7
        _check_tables(za, z, toks, f"stress {seed}")
    assert deep >= 10


def _fib_bytes(nsym, seed):
    """Bytes whose counts are Fibonacci numbers, shuffled: frequent symbols vanish into matches, the rare ones stay
    literals, and both codes of the block come out deeper than 15 bits."""
    rng = random.Random(seed)
    b = []
    for s, c in enumerate(FIB[:nsym]):
        b += [s * 7 % 251] * c
    rng.shuffle(b)
    return bytes(b)


def _stress_bytes(seed):
    rng = random.Random(seed)
    b = []
    for s in range(256):
        b += [s] * rng.choice([0, 1, 1, 2, 4, 9, 30, 200, 3000])
    rng.shuffle(b)
    return bytes(b)


def test_streams_whose_codes_need_the_length_limit(za, z):
    """k_huffman inside zg_compress: a 1 MiB chunk whose literal/length AND distance codes exceed 15 bits (the case
    the crate panics on, src/compress.rs:302-310), as one block and cut into several; blocks whose code-length code
    exceeds 7 bits (:421-426).  Byte for byte the oracle's build-mode stream; system zlib inflates it."""
    data = _fib_bytes(28, 2)  # 832 039 bytes: one chunk of 1 MiB
    toks = z.lz77_zlib(data, fast=True)
    ll, d = _hist(z, toks)
    assert _depth(ll) > 15 and _depth(d) > 15, "the input no longer drives both codes past 15 bits"
    got = za.compress(data, za.Strategy.Dynamic)
    assert got == z.compress_chunked(data, z.DYNAMIC, 1 << 20, fast=True)
    assert sysz.decompress(got) == data
    lt, dt = za.HuffmanTree.from_lz77(toks, za.zlib_compressor())
    assert lt.max_code_len() == 15 and dt.max_code_len() == 15
    # cut into 256 KiB chunks: the later chunks see the window, their histograms differ, some still exceed the limit
    got = za.compress(data, za.Strategy.Dynamic, chunk_bytes=1 << 18)
    assert got == z.compress_chunked(data, z.DYNAMIC, 1 << 18, fast=True)
    assert sysz.decompress(got) == data
    assert any(_depth(_hist(z, z.chunk_tokens(data, 1 << 18, c, fast=True))[0]) > 15 for c in range(4))
    # a second, longer corpus in two chunks of 1 MiB
    data2 = _fib_bytes(29, 5)
    got = za.compress(data2, za.Strategy.Dynamic)
    assert got == z.compress_chunked(data2, z.DYNAMIC, 1 << 20, fast=True)
    assert sysz.decompress(got) == data2
    # code-length codes deeper than 7
    deep = 0
    for seed in (0, 3, 4, 5, 7, 8, 10, 11, 13, 14):
        blob = _stress_bytes(seed)
        deep += _cl_depth(z, z.lz77_zlib(blob, fast=True)) > 7
        got = za.compress(blob, za.Strategy.Dynamic)
        assert got == z.compress_chunked(blob, z.DYNAMIC, 1 << 20, fast=True), f"stress bytes {seed}"
        assert sysz.decompress(got) == blob
    assert deep >= 5, "the inputs no longer drive the code-length code past 7 bits"


def test_record_batch_with_codes_beyond_the_limits(za, z):
    """The same inputs as records of a batch (every record a stream of its own), beside records with a lone distance
    symbol and an empty one."""
    recs = [_fib_bytes(24, 3), _fib_bytes(26, 4), _stress_bytes(0), _stress_bytes(3), b"hello hello hello hello", b"",
            _fib_bytes(28, 1)[:300_000]]
    got = za.compress_batch(recs, za.Strategy.Dynamic)
    for i, (r, g) in enumerate(zip(recs, got)):
        assert g == z.compress_chunked(r, z.DYNAMIC, 1 << 20, fast=True), f"record {i}"
        assert sysz.decompress(g) == r
    assert _depth(_hist(z, z.lz77_zlib(recs[1], fast=True))[0]) > 15
