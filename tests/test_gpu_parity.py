"""Parity of the HIP path (through the C ABI) with the CPU oracle."""
import random
import zlib as sysz

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def za():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    torch.zeros(1, device="cuda")  # bring torch's HIP context up before the library's
    import zlib_amd
    zlib_amd.default_context()
    return zlib_amd


@pytest.fixture(scope="module")
def z():
    from oracle import zref
    zref.lib()
    return zref


def _same(got, want, what=""):
    """Equality of two long sequences; on a mismatch reports the first differing index (pytest's own diff of
    sequences this long takes minutes)."""
    if got == want:
        return
    n = min(len(got), len(want))
    i = next((k for k in range(n) if got[k] != want[k]), n)
    raise AssertionError(f"{what}: lengths {len(got)} / {len(want)}, first difference at {i}: "
                         f"{list(got[i:i + 4])} != {list(want[i:i + 4])}")


def _small_cases():
    rng = random.Random(1234)
    cases = [b"", b"a", b"ab", b"abc", b"abcd", b"aaaa", b"aaaaa", b"a" * 300, b"a" * 1200,
             b"abcabcabcabcabcabcabcabc_abcabc", b"ab" * 700, bytes(range(256)) * 3, b"hello hello hello hello"]
    for n in (50, 400, 3000):
        cases.append(bytes(rng.randrange(256) for _ in range(n)))
        cases.append(bytes(rng.randrange(2) for _ in range(n)))
        cases.append(bytes(rng.choice(b"abc") for _ in range(n)))
    return cases


def test_adler32(za, z):
    rng = random.Random(3)
    for n in (0, 1, 15, 16, 17, 5552, 70000, 300000, 1 << 20):
        data = bytes(rng.randrange(256) for _ in range(min(n, 4096))) * (n // 4096 + 1)
        data = data[:n]
        assert za.adler32(data) == z.adler32(data) == sysz.adler32(data)
    assert za.adler32(b"\xff" * 3_000_000) == sysz.adler32(b"\xff" * 3_000_000)


def test_lz77_tokens_small(za, z):
    comp = za.zlib_compressor()
    for data in _small_cases():
        _same(comp.compress_packed(data), z.lz77_zlib(data, fast=True), data[:32])
        if len(data) > 10:
            for start in (1, len(data) // 2, len(data) - 2):
                _same(comp.compress_packed(data, start=start), z.lz77_zlib(data, start, fast=True), "")


def test_lz77_default_parameters(za, z):
    """LZ77Compressor::new() uses window 144, min 5, max 100 (src/lz77.rs:18-21)."""
    comp = za.LZ77Compressor.new()
    rng = random.Random(8)
    cases = [b"ABCBDBDBDBDBDBDBCBADBSB", b"hi hi hello hi hi wow hello hi", b"abcdefabcdefabcdef",
             b"abcdefghijklmnopqrstuvwxyz" * 100, b"a" * 1000, bytes(range(255))]
    cases += [bytes(rng.choice(b"abcd") for _ in range(2000))]
    for data in cases:
        _same(comp.compress_packed(data), z.lz77_compress(data), data[:32])
    small, large = za.LZ77Compressor.with_window_size(64), za.LZ77Compressor.with_window_size(1024)
    inp = b"abcdefghijklmnopqrstuvwxyz" * 100
    _same(small.compress_packed(inp), z.lz77_compress(inp, window=64), "")
    _same(large.compress_packed(inp), z.lz77_compress(inp, window=1024), "")
    assert len(small.compress(inp)) > len(large.compress(inp))  # lz77.rs:317-327


def test_lz77_tokens_text_and_binary(za, z):
    comp = za.zlib_compressor()
    for kind, n in (("text", 200_000), ("binary", 100_000), ("random", 70_000), ("mixed", 150_000)):
        data = za.corpus(kind, n)
        _same(comp.compress_packed(data), z.lz77_zlib(data, fast=True), kind)


@pytest.mark.parametrize("window,min_len,max_len", [(2048, 3, 258), (4096, 4, 258), (4096, 3, 64), (8192, 6, 258),
                                                    (20000, 3, 258), (32737, 3, 258), (32750, 4, 100),
                                                    (32768, 5, 258), (32768, 4, 258)])
def test_lz77_parameter_sweep(za, z, window, min_len, max_len):
    """Windows on both sides of the index kernels' limits and every min_match_length class of the two-level
    search (3, 4, 5 and above), against the verbatim candidate loop."""
    comp = za.LZ77Compressor.with_window_size(window)
    comp.min_match_length, comp.max_match_length = min_len, max_len
    rng = random.Random(window * 7 + min_len)
    cases = [za.corpus("text", 90_000, seed=window), za.corpus("binary", 70_000, seed=min_len),
             bytes(rng.choice(b"abcde") for _ in range(50_000)),
             (b"0123456789abcdef" * 40 + bytes(rng.randrange(256) for _ in range(37))) * 90]
    for data in cases:
        want = z.lz77_compress(data, window=window, min_len=min_len, max_len=max_len)
        _same(comp.compress_packed(data), want, (window, min_len, max_len, data[:16]))


def test_far_members_at_the_edge_of_the_window(za, z):
    """Chains whose links are almost a full window long, so that the entry the index kernel records lies up to
    two windows behind the position."""
    rng = random.Random(99)
    marker = b"\xfe\xfdQZ#@!"
    for gap in (32768, 32767, 32766, 32705, 32704, 30000):
        filler = lambda n: bytes(rng.randrange(64) for _ in range(n))
        data = b"".join(marker + filler(gap - len(marker)) for _ in range(5)) + marker + filler(100)
        _same(za.zlib_compressor().compress_packed(data), z.lz77_zlib(data, fast=True), gap)


def test_long_three_byte_chains(za, z):
    """Positions whose first three bytes are everywhere in the window but whose fourth is rare: the level-2 walk
    runs out of its step budget and a whole wave finishes it.  Includes four-byte agreements at distances 4, 5
    and 6 (the first two end the search instead of being recorded)."""
    rng = random.Random(5)
    alphabet = b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz"
    parts = []
    for i in range(9000):
        parts.append(b"xyz" + bytes([rng.choice(alphabet)]) + bytes([rng.randrange(256)]))
        if i % 1500 == 7:
            parts.append(b"xyz~" + b"xyz~"[: i % 3 + 1] + b"xyz~Q")  # near repeats right behind a rare fourth byte
    data = b"".join(parts)
    comp = za.zlib_compressor()
    _same(comp.compress_packed(data), z.lz77_zlib(data, fast=True), "xyz chains")
    comp.min_match_length = 4
    _same(comp.compress_packed(data), z.lz77_compress(data, window=32768, min_len=4, max_len=258), "min 4")
    got = za.compress(data, za.Strategy.Dynamic, chunk_bytes=16384)
    _same(got, z.compress_chunked(data, z.DYNAMIC, 16384, fast=True), "stream")


def test_conservative_index_kernel(za, z, monkeypatch):
    """The index variant the library falls back to when the exchange-order check trips gives the same streams."""
    monkeypatch.setenv("ZLIB_AMD_SAFE_INDEX", "1")
    ctx = za.Context(0)
    try:
        for data in (za.corpus("text", 400_000), za.corpus("mixed", 300_000, seed=5), b"a" * 70_000):
            got = za.compress(data, za.Strategy.Dynamic, chunk_bytes=65536, ctx=ctx)
            _same(got, za.compress(data, za.Strategy.Dynamic, chunk_bytes=65536), "")
            _same(got, z.compress_chunked(data, z.DYNAMIC, 65536, fast=True), "")
        # record batches fall back as well (isolated slots: the direct scan that stays inside a slot)
        recs = [za.corpus("records", 4096, offset=4096 * i) for i in range(40)] + [b"", b"abcabcabcabc" * 50,
                                                                                   za.corpus("text", 9000, seed=3)]
        got = za.compress_batch(recs, za.Strategy.Dynamic, ctx=ctx)
        assert got == za.compress_batch(recs, za.Strategy.Dynamic)
        for r, g in zip(recs, got):
            _same(g, z.compress_chunked(r, z.DYNAMIC, 1 << 20, fast=True), f"record of {len(r)} bytes")
    finally:
        ctx.close()


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0], [0] * 5])
def test_several_devices_in_one_process_give_the_single_device_stream(za, z, devices):
    """zg_compress_multi with the one GPU of the box standing in for several (a context per run): the assembled
    stream is the oracle's stream for the whole input (and the single-context stream), whatever the bit phase at
    which the runs meet."""
    for data, chunk in ((za.corpus("mixed", (5 << 20) + 12345, seed=21), 1 << 16), (za.corpus("text", 700_001, seed=22), 1 << 17),
                        (za.corpus("text", 100_000, seed=23), 1 << 20), (b"", 1 << 20), (za.corpus("binary", 3 << 16, seed=24), 1 << 16)):
        for strategy in (za.Strategy.Dynamic, za.Strategy.Fixed):
            got = za.compress_multi(data, devices, strategy, chunk_bytes=chunk)
            zs = z.DYNAMIC if strategy == za.Strategy.Dynamic else z.FIXED
            _same(got, z.compress_chunked(data, zs, chunk, fast=True), f"oracle: {len(data)} bytes over {len(devices)} runs")
            _same(got, za.compress(data, strategy, chunk_bytes=chunk), f"{len(data)} bytes over {len(devices)} runs")
            assert sysz.decompress(got) == data
    assert za.compress_multi(b"abc" * 5000, devices, za.Strategy.Raw) == za.compress(b"abc" * 5000, za.Strategy.Raw)
    assert sysz.decompress(za.compress_multi(za.corpus("text", 300_000), devices, za.Strategy.Auto)) == za.corpus("text", 300_000)


def test_search_along_the_parse(za, z, monkeypatch):
    """The optional search that visits only the positions the greedy parse can reach (k_walk; four-byte chains
    in LDS, three-byte chains beside them) gives the streams and units of the full search."""
    monkeypatch.setenv("ZLIB_AMD_WALK", "1")
    ctx = za.Context(0)
    try:
        rng = random.Random(77)
        cases = [(za.corpus("text", 3 << 20, seed=11), 1 << 20), (za.corpus("mixed", (4 << 20) + 777, seed=12), 1 << 20),
                 (za.corpus("binary", 1 << 20, seed=13), 1 << 18), (za.corpus("random", 300_000, seed=14), 65536),
                 (b"a" * 200_000, 65536), (b"abcdefgh" * 30_000, 1 << 20), (bytes(300_000), 1 << 17),
                 (bytes(rng.choice(b"ab") for _ in range(150_000)), 32768),
                 (b"".join(bytes([rng.randrange(4)]) * rng.randint(1, 600) for _ in range(2000)), 65536),
                 (za.corpus("text", 40_000, seed=15), 1 << 20), (b"", 1024), (b"abc", 1024)]
        for data, chunk in cases:
            got = za.compress(data, za.Strategy.Dynamic, chunk_bytes=chunk, ctx=ctx)
            _same(got, za.compress(data, za.Strategy.Dynamic, chunk_bytes=chunk), f"{len(data)} bytes, chunk {chunk}")
        # units with other parameters: windows and minimum lengths (4: no three-byte chains; 5 and more)
        data = za.corpus("text", 90_000, seed=16) + bytes(rng.choice(b"xyz") for _ in range(30_000))
        for window, mn, mx in ((32768, 3, 258), (32768, 4, 258), (8192, 5, 100), (4096, 3, 32), (2048, 6, 258)):
            comp = za.LZ77Compressor.with_window_size(window)
            comp.min_match_length, comp.max_match_length = mn, mx
            _same(comp.compress_packed(data[:20_000], ctx=ctx), z.lz77_compress(data[:20_000], 0, window, mn, mx),
                  f"w={window} min={mn} max={mx}")
            _same(comp.compress_packed(data, ctx=ctx), comp.compress_packed(data), f"w={window} min={mn} max={mx} long")
        # a batch of large records (isolated slots with the general tile geometry)
        recs = [za.corpus("text", 70_000, seed=1), za.corpus("binary", 50_000, seed=2), b"", za.corpus("mixed", 33_000, seed=4)]
        assert za.compress_batch(recs, ctx=ctx) == za.compress_batch(recs)
    finally:
        ctx.close()


def test_limits_are_refused_not_wrapped(za):
    """Chunks and record slots above 256 MiB would overflow 32-bit bit offsets inside a block: refused."""
    import ctypes as C
    import torch
    from zlib_amd import _lib
    L = _lib.lib()
    ctx = za.default_context()
    d = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    out_n = C.c_size_t()
    o = _lib.Opts((256 << 20) + 1024, 0)
    assert L.zg_compress_bound(10, 1, C.byref(o)) == 0
    assert L.zg_compress_device(ctx.handle, d.data_ptr(), 1024, 1, C.byref(o), d.data_ptr(), 4096,
                                C.byref(out_n)) == _lib.ERR_UNSUPPORTED
    assert L.zg_compress_batch_device(ctx.handle, d.data_ptr(), (256 << 20) + 1024, None, 1, 1, d.data_ptr(), 4096,
                                      d.data_ptr(), C.byref(out_n)) == _lib.ERR_UNSUPPORTED
    o = _lib.Opts(256 << 20, 0)  # the accepted maximum
    assert L.zg_compress_bound(10, 1, C.byref(o)) > 0
    # zg_decompress_batch_device: an output buffer without offsets, streams without input
    assert L.zg_decompress_batch_device(ctx.handle, d.data_ptr(), d.data_ptr(), 1, d.data_ptr(), None, d.data_ptr(),
                                        d.data_ptr()) == _lib.ERR_ARG
    assert L.zg_decompress_batch_device(ctx.handle, None, d.data_ptr(), 1, None, None, d.data_ptr(),
                                        d.data_ptr()) == _lib.ERR_ARG


def test_huffman_tables(za, z):
    rng = random.Random(4)
    for data in [za.corpus("text", 50_000), za.corpus("binary", 30_000), b"a" * 5000, b"hello hello hello hello", b""]:
        toks = z.lz77_zlib(data, fast=True)
        lt, dt = za.HuffmanTree.from_lz77(toks, za.zlib_compressor())
        ll_len, ll_code, d_len, d_code = z.chunk_tables(toks)
        assert lt.encodings() == {s: (ll_code[s], ll_len[s]) for s in range(286) if ll_len[s]}
        assert dt.encodings() == {s: (d_code[s], d_len[s]) for s in range(30) if d_len[s]}


@pytest.mark.parametrize("strategy", ["Dynamic", "Fixed", "Raw", "Auto"])
def test_streams_small(za, z, strategy):
    st, zs = getattr(za.Strategy, strategy), getattr(z, strategy.upper())
    for data in _small_cases():
        for chunk in (1 << 20, 1024):
            got = za.compress(data, st, chunk_bytes=chunk)
            _same(got, z.compress_chunked(data, zs, chunk, fast=True), (data[:32], chunk))
            assert sysz.decompress(got) == data


def test_auto_reaches_every_branch_of_the_heuristic(za, z):
    """Strategy::Auto (src/compress.rs:89-169): below 256 bytes stored blocks, below 1 KiB the fixed code, from there
    on the first KiB decides -- entropy against log2 of its alphabet (:125-138), the share of its five RAREST symbols
    (:140-150, sorted ascending), and, when the two cancel, symbols x log2(len) / len (:157-167).  One input per
    branch and threshold; the expected block type is what the restated heuristic yields, the stream must equal the
    oracle's Auto stream and that strategy's own stream."""
    rng = random.Random(2026)
    text = za.corpus("text", 6000)
    skew = bytes(rng.choice(b"a" * 99 + b"b") for _ in range(4000))
    dominant = bytearray(b"a" * 1024)  # one symbol and 51 single ones: proximity 0.1, the third rule decides
    for i in range(51):
        dominant[17 * i + 3] = 100 + i
    cases = [
        (bytes(rng.randrange(256) for _ in range(255)), "Raw"),       # :91
        (bytes(rng.randrange(256) for _ in range(256)), "Fixed"),     # :97
        (text[:1023], "Fixed"),
        (text[:1024], "Fixed"),                                         # entropy close to log2(alphabet): :152
        (text, "Fixed"),
        (bytes(rng.randrange(256) for _ in range(5000)), "Fixed"),
        (skew[:3000], "Dynamic"),                                       # five symbols hold everything: :154
        (bytes(rng.choice(b"ab") for _ in range(3000)), "Dynamic"),   # the two cancel, ratio 0.008: :163
        (bytes(3000), "Dynamic"),                                       # one symbol: 0 / 0, every comparison false
        (bytes(dominant) + b"a" * 16, "Fixed"),                         # the two cancel, 52 * log2(1040) / 1040 > 0.5: :165
        (bytes(dominant) + b"a" * 176, "Dynamic"),                      # ... 52 * log2(1200) / 1200 < 0.5
        (bytes(rng.randrange(256) for _ in range(1024)) + text[:3000], "Fixed"),   # only the first KiB is looked at
        (skew[:1024] + bytes(rng.randrange(256) for _ in range(3000)), "Dynamic"),
    ]
    for data, want in cases:
        got = za.compress(data, za.Strategy.Auto)
        _same(got, z.compress_chunked(data, z.AUTO, 1 << 20, fast=True), (len(data), want))
        _same(got, za.compress(data, getattr(za.Strategy, want)), (len(data), want))
        assert sysz.decompress(got) == data


@pytest.mark.parametrize("strategy", ["Dynamic", "Fixed"])
def test_streams_equal_reference_single_chunk(za, z, strategy):
    """n <= chunk_bytes: the stream is the reference crate's stream byte for byte."""
    st, zs = getattr(za.Strategy, strategy), getattr(z, strategy.upper())
    for data in [za.corpus("text", 30_000), za.corpus("text", 3000, seed=7), b"a" * 1200, b"abcd"]:
        _same(za.compress(data, st), z.compress_reference(data, zs), "")


@pytest.mark.parametrize("kind,n,chunk", [("text", 300_000, 65536), ("mixed", 3 << 20, 1 << 20),
                                          ("binary", 200_000, 32768), ("random", 100_000, 16384),
                                          ("text", 1_000_000, 1 << 20), ("records", 40_960, 4096)])
def test_streams_corpora(za, z, kind, n, chunk):
    data = za.corpus(kind, n)
    got = za.compress(data, za.Strategy.Dynamic, chunk_bytes=chunk)
    assert sysz.decompress(got) == data
    _same(got, z.compress_chunked(data, z.DYNAMIC, chunk, fast=True), "")
    assert z.decompress(got) == data


def test_long_runs_and_periodic_data(za, z):
    """Inputs whose parse units are long: the speculative parse cannot re-synchronise inside a sub-tile and the
    serial re-walk has to take over; the match search hits its early-outs."""
    rng = random.Random(17)
    rec = bytes(rng.randrange(256) for _ in range(16))
    cases = [bytes(70_000), b"\x07" * 33_000 + bytes(40_000), rec * 5000, (b"ab" * 600 + b"c") * 60,
             bytes(rng.randrange(2) for _ in range(60_000)), b"abcdefgh" * 9000 + bytes(rng.randrange(256) for _ in range(5000))]
    comp = za.zlib_compressor()
    for data in cases:
        _same(comp.compress_packed(data), z.lz77_zlib(data, fast=True), "")
        for chunk in (1 << 20, 16384):
            got = za.compress(data, za.Strategy.Dynamic, chunk_bytes=chunk)
            _same(got, z.compress_chunked(data, z.DYNAMIC, chunk, fast=True), "")
            assert sysz.decompress(got) == data


def test_mostly_incompressible_tiles_with_sparse_repeats(za, z):
    """Random bytes with a few short strings sprinkled in: level 1 leaves nearly every position open, so level 2 takes
    the three-byte chains first and looks for four-byte candidates along them; the sprinkled strings recur often
    enough for their chains to outrun a walk's budget (parked scans, the four-byte pass after all), at distances
    around the limits of five and of the strings' own lengths."""
    rng = random.Random(515)
    marks = [b"ABCD", b"XYZ", b"QRSTU", b"ABCE", b"XYZW"]
    for n, gap in ((300_000, 150), (200_000, 40), (120_000, 9)):
        out = bytearray()
        while len(out) < n:
            out += bytes(rng.randrange(256) for _ in range(rng.randint(1, 2 * gap)))
            out += rng.choice(marks)
            if rng.random() < 0.1:
                out += rng.choice(marks)  # back to back: distances of three to five
        data = bytes(out[:n])
        _same(za.zlib_compressor().compress_packed(data), z.lz77_zlib(data, fast=True), (n, gap))
        for chunk in (1 << 20, 65536):
            got = za.compress(data, za.Strategy.Dynamic, chunk_bytes=chunk)
            _same(got, z.compress_chunked(data, z.DYNAMIC, chunk, fast=True), (n, gap, chunk))
            assert sysz.decompress(got) == data


@pytest.mark.gpu
def test_motif_soup_long_chains_with_periodic_neighbourhoods(za, z):
    """Few symbols, short motifs repeated a random number of times: five-byte chains thousands of members long,
    best lengths that grow step by step along them, and search-ending candidates a few bytes before the
    position -- the cases in which the level-1 walk changes to a more selective chain."""
    import random
    comp = za.zlib_compressor()
    for seed in range(6):
        rng = random.Random(1000 + seed)
        motifs = [bytes(rng.choice(b"abc") for _ in range(rng.randint(1, 9))) for _ in range(12)]
        parts, n = [], 0
        while n < 90_000:
            m = rng.choice(motifs) * rng.randint(1, 6)
            if rng.random() < 0.05:
                m += bytes([rng.randrange(256)])
            parts.append(m)
            n += len(m)
        data = b"".join(parts)
        _same(comp.compress_packed(data), z.lz77_zlib(data, fast=True), f"motif soup {seed}")
    # a frequent five bytes followed by a counter: the chain of the sixth byte is short
    rec = b"".join(b"\x3f\x00\x00\x00\x00" + i.to_bytes(4, "little") + bytes([(i * 7) & 0xFF, 0, 0]) for i in range(9000))
    _same(comp.compress_packed(rec), z.lz77_zlib(rec, fast=True), "counter records")


def test_batching_does_not_change_the_stream(za):
    data = za.corpus("text", 700_000)
    a = za.compress(data, za.Strategy.Dynamic, chunk_bytes=65536)
    b = za.compress(data, za.Strategy.Dynamic, chunk_bytes=65536, batch_bytes=131072)
    _same(a, b, "")


def test_host_path_above_the_page_touch_threshold(za):
    """zg_compress on an input large enough for the result pages to be touched ahead: the stream inflates to the
    input and equals the one made batch by batch with a small batch."""
    n = (64 << 20) + 12_345
    data = za.corpus("mixed", n)
    a = za.compress(data, za.Strategy.Dynamic)
    assert sysz.decompress(a) == data
    b = za.compress(data, za.Strategy.Dynamic, batch_bytes=16 << 20)
    _same(a, b, "")


def test_raw_large(za, z):
    data = za.corpus("random", 1_000_003)
    got = za.compress(data, za.Strategy.Raw)
    _same(got, z.compress_chunked(data, z.RAW, 1 << 20), "")
    assert sysz.decompress(got) == data


@pytest.mark.parametrize("world,total,chunk", [(2, 300_000, 65536), (3, 200_000, 16384), (4, 4 << 20, 1 << 20),
                                               (3, 100_000, 65536)])  # the last: two chunks, the first rank without input
def test_shard_api_assembles_the_single_gpu_stream(za, z, world, total, chunk):
    """zg_shard_begin / zg_shard_finish for every rank's run of chunks (run one after the other on this GPU),
    assembled with the host logic the multi-GPU bench uses, give the stream of a single compress call."""
    import torch
    from zlib_amd import sharding
    data = za.corpus("mixed" if total > (1 << 20) else "text", total, seed=21)
    whole = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    plan = sharding.plan_shards(total, chunk, world)
    engines, infos = [], []
    for r, (start, n, halo) in enumerate(plan):
        ctx = za.Context(0)
        view = whole[start - halo:start + n].clone()  # the rank's own copy: halo bytes, then its run
        out = torch.zeros(za.compress_bound(n, za.Strategy.Dynamic, chunk_bytes=chunk) + 8, dtype=torch.uint8, device="cuda")
        eng = sharding.HipEngine(ctx, view.data_ptr(), halo, n, za.Strategy.Dynamic, chunk, out)
        eng._keep = view
        engines.append(eng)
        infos.append(eng.begin(r == world - 1))
    bases, end_bit, adler = sharding.stream_layout(infos)
    total_bytes = (end_bit + 7) // 8 + 4
    final = bytearray(total_bytes)
    for r, eng in enumerate(engines):
        nb = eng.finish(bases[r], r == 0, r == world - 1, adler)
        piece = eng.out[:nb].cpu().numpy().tobytes()
        off = 0 if r == 0 else bases[r] // 8
        for i, v in enumerate(piece[:1]):
            final[off + i] |= v
        final[off + 1:off + nb] = piece[1:]
    _same(bytes(final), z.compress_chunked(data, z.DYNAMIC, chunk, fast=True), "oracle")
    _same(bytes(final), za.compress(data, za.Strategy.Dynamic, chunk_bytes=chunk), "")
    assert sysz.decompress(bytes(final)) == data


@pytest.mark.parametrize("world,total,chunk,batch", [(3, (6 << 20) + 4321, 1 << 16, 1 << 19), (2, 300_000, 16384, 16384),
                                                     (4, 3 << 20, 1 << 18, 1 << 20), (1, 1_000_000, 1 << 17, 1 << 18),
                                                     (3, 100_000, 65536, 65536), (2, 0, 65536, 65536)])  # ranks without input
def test_shard_pieces_assemble_the_single_gpu_stream(za, z, world, total, chunk, batch):
    """zg_shard_open / zg_shard_next for every rank's run (one after the other on this GPU), pass by pass; every
    piece is copied away the moment it is reported, as an exchange that runs beside the kernels would, and the
    copies -- not the rank's buffer as it is at the end -- are merged (zg_shard_merge) and sealed (zg_shard_seal)
    on the first rank's context.  The result is the oracle's stream."""
    import torch
    from zlib_amd import sharding
    data = za.corpus("mixed" if total > (1 << 20) else "text", total, seed=31)
    whole = torch.frombuffer(bytearray(data) or bytearray(1), dtype=torch.uint8).cuda()
    plan = sharding.plan_shards(total, chunk, world)
    bound = lambda n: za.compress_bound(n, za.Strategy.Dynamic, chunk_bytes=chunk)
    stages, infos, n_pieces, ctxs = [], [], [], []
    for r, (start, n, halo) in enumerate(plan):
        ctx = za.Context(0)
        ctxs.append(ctx)
        view = whole[start - halo:start + n].clone()
        packed = torch.full((bound(n) + 8,), 0xEE, dtype=torch.uint8, device="cuda")
        stage = torch.full((sharding.stage_bytes(n, bound),), 0xEE, dtype=torch.uint8, device="cuda")
        eng = sharding.HipPieceEngine(ctx, view.data_ptr(), halo, n, za.Strategy.Dynamic, chunk, batch, packed)
        eng.open(r == world - 1)
        at = k = 0
        while True:
            off, nb, bits, last, ad = eng.next()
            assert off == at and (bits // 8 == off + nb if not last else (bits + 7) // 8 == off + nb)
            stage[off:off + nb].copy_(packed[off:off + nb])
            at, k = at + nb, k + 1
            if last:
                break
        n_blocks = -(-n // chunk) or (1 if r == world - 1 else 0)  # a run without bytes adds nothing unless it is the last
        assert k == max(1, -(-n_blocks // max(1, batch // chunk)))
        torch.cuda.synchronize()
        assert torch.equal(stage[:at], packed[:at]), "a piece changed after it had been reported"
        n_pieces.append(k)
        stages.append(stage)
        infos.append((bits, ad, n))
        blocks = za.last_block_offsets(ctx=ctx)
        assert len(blocks) == n_blocks and blocks[:1] in ([], [0]) and all(a < b for a, b in zip(blocks, blocks[1:]))
        with pytest.raises(za.ZgError):
            eng.next()  # the run has ended
    bases, end_bit, adler = sharding.stream_layout(infos)
    final = torch.full(((end_bit + 7) // 8 + 4 + 16,), 0xEE, dtype=torch.uint8, device="cuda")
    head = sharding.HipPieceEngine(ctxs[0], 0, 0, 0, za.Strategy.Dynamic, chunk, batch, None)
    for r in range(world):
        head.merge(stages[r], infos[r][0], final, bases[r], r == 0)
    nb = head.seal(final, end_bit, adler)
    got = final[:nb].cpu().numpy().tobytes()
    _same(got, z.compress_chunked(data, z.DYNAMIC, chunk, fast=True), f"oracle, pieces per rank {n_pieces}")
    assert sysz.decompress(got) == data
    for c in ctxs:
        c.close()


def test_shard_pieces_through_the_assembly_logic_and_refusals(za, z):
    """One rank through sharding.compress_sharded_overlapped (the calls a multi-GPU run makes, nothing to exchange);
    what the piecewise calls refuse."""
    import torch
    from zlib_amd import sharding
    total, chunk = (2 << 20) + 999, 1 << 17
    data = za.corpus("mixed", total, seed=33)
    d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    cap = za.compress_bound(total, za.Strategy.Dynamic, chunk_bytes=chunk) + 8
    packed = torch.empty(cap, dtype=torch.uint8, device="cuda")
    final = torch.empty(cap, dtype=torch.uint8, device="cuda")
    ctx = za.Context(0)
    eng = sharding.HipPieceEngine(ctx, d_in.data_ptr(), 0, total, za.Strategy.Dynamic, chunk, 1 << 19, packed)
    for _ in range(2):
        n = sharding.compress_sharded_overlapped(eng, sharding.SingleRank, torch, 0, 1, torch.device("cuda"),
                                                 final=final, stages=[None])
        _same(final[:n].cpu().numpy().tobytes(), z.compress_chunked(data, z.DYNAMIC, chunk, fast=True), "oracle")
    with pytest.raises(za.ZgError):  # no run is open
        za.shard_next(ctx=ctx)
    with pytest.raises(za.ZgError):  # the packed buffer is too small
        za.shard_open(d_in.data_ptr(), 0, total, za.Strategy.Dynamic, True, packed.data_ptr(), 1000, chunk_bytes=chunk, ctx=ctx)
    with pytest.raises(za.ZgError):  # the first run begins behind the two header bytes
        za.shard_merge(packed.data_ptr(), 100, final.data_ptr(), cap, 24, True, ctx=ctx)
    with pytest.raises(za.ZgError):  # no room
        za.shard_merge(packed.data_ptr(), 8 * 4096, final.data_ptr(), 4096, 16, True, ctx=ctx)
    with pytest.raises(za.ZgError):
        za.shard_seal(final.data_ptr(), 100, 8 * 100, 1, ctx=ctx)
    # a whole-stream call on the same context still works afterwards
    _same(za.compress(data, za.Strategy.Dynamic, chunk_bytes=chunk, ctx=ctx),
          z.compress_chunked(data, z.DYNAMIC, chunk, fast=True), "after the refusals")
    ctx.close()


def test_several_devices_failing_call_leaves_nothing_behind(za, z):
    """zg_compress_multi with a device that does not exist: the call fails as a whole, and the next one works."""
    data = za.corpus("text", 400_000, seed=35)
    with pytest.raises(za.ZgError):
        za.compress_multi(data, [0, 4096], za.Strategy.Dynamic, chunk_bytes=1 << 16)
    _same(za.compress_multi(data, [0, 0], za.Strategy.Dynamic, chunk_bytes=1 << 16),
          z.compress_chunked(data, z.DYNAMIC, 1 << 16, fast=True), "after a failed call")


def _record_cases(za):
    rng = random.Random(77)
    recs = [b"", b"a", b"abc", b"hello hello hello hello", bytes(4096), b"\x01" * 1000,
            bytes(rng.randrange(256) for _ in range(4096)), b"one repeated phrase " * 200,
            bytes(rng.randrange(256) for _ in range(777))]
    recs += [za.corpus("records", 4096, offset=4096 * i) for i in range(12)]
    recs += [za.corpus("text", n, seed=n) for n in (1, 100, 1023, 1024, 1025, 3000, 4095, 5000)]
    recs += [za.corpus("binary", 2500, seed=3), za.corpus("mixed", 4096, seed=9)]
    return recs


@pytest.mark.parametrize("strategy", ["Dynamic", "Fixed"])
def test_record_batch_equals_per_record_compress(za, z, strategy):
    """zg_compress_batch: every record's stream is compress(record, strategy) -- the reference's single-block
    stream wherever the reference does not panic -- and records do not see each other."""
    st, zs = getattr(za.Strategy, strategy), getattr(z, strategy.upper())
    recs = _record_cases(za)
    got = za.compress_batch(recs, st)
    assert len(got) == len(recs)
    for i, (r, g) in enumerate(zip(recs, got)):
        _same(g, z.compress_chunked(r, zs, 1 << 20, fast=True), f"record {i} ({len(r)} bytes)")
        assert sysz.decompress(g) == r
    # the same records in another order and as equally sized slots (the copy-free layout)
    got2 = za.compress_batch(recs[::-1], st)
    assert got2 == got[::-1]
    same = [za.corpus("records", 4096, offset=4096 * i) for i in range(300)]
    for r, g in zip(same, za.compress_batch(same, st)):
        _same(g, z.compress_chunked(r, zs, 1 << 20, fast=True), "equal slots")


def test_record_batch_large_records_and_device_entry(za, z):
    import torch
    recs = [za.corpus("text", 70_000, seed=1), za.corpus("mixed", 40_000, seed=2), b"", za.corpus("text", 33_000, seed=4)]
    for r, g in zip(recs, za.compress_batch(recs)):
        _same(g, z.compress_chunked(r, z.DYNAMIC, 1 << 20, fast=True), len(r))
    # device entry point, 5000 records of 4 KiB
    n, stride = 5000, 4096
    data = za.corpus("records", n * stride)
    d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    cap = za.compress_batch_bound(stride, n)
    d_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    d_off = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    total = za.compress_batch_device(d_in.data_ptr(), stride, 0, n, d_out.data_ptr(), cap, d_off.data_ptr())
    off = d_off.cpu().tolist()
    out = d_out[:total].cpu().numpy().tobytes()
    assert off[0] == 0 and off[-1] == total
    for i in range(0, n, 97):
        rec = data[i * stride:(i + 1) * stride]
        assert sysz.decompress(out[off[i]:off[i + 1]]) == rec
        _same(out[off[i]:off[i + 1]], z.compress_chunked(rec, z.DYNAMIC, 1 << 20, fast=True), i)
    assert sysz.decompress(out[off[n - 1]:off[n]]) == data[(n - 1) * stride:]


def test_record_batch_in_many_passes_and_under_the_phase_timers(za, z, monkeypatch):
    """Passes of 1 MiB (256 records of 4 KiB, 777 of them in 4 passes; 1300 of 1 KiB in 2): the streams' running byte
    offset changes words from pass to pass and the checksums of a pass are taken on a second stream beside its bit
    packing, with or without the phase timers' events around every launch.  The result is the same either way,
    and every stream is the oracle's."""
    import torch
    monkeypatch.setenv("ZLIB_AMD_BATCH_PASS_MIB", "1")
    ctx = za.default_context()
    for n, stride in ((777, 4096), (1300, 1024), (256, 4096), (257, 4096)):
        data = za.corpus("records", n * stride, seed=n)
        d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        cap = za.compress_batch_bound(stride, n)
        outs = []
        for timers in (False, True, False):
            ctx.profile(timers)
            try:
                d_out = torch.full((cap,), 0x55, dtype=torch.uint8, device="cuda")
                d_off = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
                total = za.compress_batch_device(d_in.data_ptr(), stride, 0, n, d_out.data_ptr(), cap, d_off.data_ptr(), ctx=ctx)
            finally:
                ctx.profile(False)
            off = d_off.cpu().tolist()
            assert off[0] == 0 and off[n] == total and all(a < b for a, b in zip(off, off[1:]))
            outs.append((off, d_out[:total].cpu().numpy().tobytes()))
        assert outs[0] == outs[1] == outs[2]
        off, out = outs[0]
        for i in list(range(0, n, 41)) + [255, n - 1]:
            rec = data[i * stride:(i + 1) * stride]
            _same(out[off[i]:off[i + 1]], z.compress_chunked(rec, z.DYNAMIC, 1 << 20, fast=True), (n, stride, i))


@pytest.mark.parametrize("strategy", ["Dynamic", "Fixed"])
def test_record_batch_into_a_dirty_buffer(za, z, strategy):
    """The record path clears only the words its bit packer ORs into, not the output: every byte of every stream must
    come out right over whatever the buffer held -- two batches of different records into the same buffer of 0xEE
    bytes, the second one shorter; records of odd lengths so that streams begin at every byte phase; records of 4 KiB
    over more than one pass."""
    import torch
    st, zs = getattr(za.Strategy, strategy), (z.DYNAMIC if strategy == "Dynamic" else z.FIXED)
    rng = random.Random(5)
    stride = 1024  # (slots are multiples of 1 KiB; the records in them are not)
    for n, seed in ((700, 1), (300, 2)):
        recs = [za.corpus("records" if i % 3 else "text", rng.randint(0, stride), seed=seed, offset=997 * i) for i in range(n)]
        slots = b"".join(r + bytes(stride - len(r)) for r in recs)
        d_in = torch.frombuffer(bytearray(slots), dtype=torch.uint8).cuda()
        d_len = torch.tensor([len(r) for r in recs], dtype=torch.int32).cuda()
        cap = za.compress_batch_bound(stride, 700, st)
        if n == 700:
            d_out = torch.full((cap,), 0xEE, dtype=torch.uint8, device="cuda")
            d_off = torch.zeros(701, dtype=torch.int64, device="cuda")
        total = za.compress_batch_device(d_in.data_ptr(), stride, d_len.data_ptr(), n, d_out.data_ptr(), cap, d_off.data_ptr(), st)
        off = d_off.cpu().tolist()
        out = d_out[:total].cpu().numpy().tobytes()
        assert off[0] == 0 and off[n] == total
        for i, r in enumerate(recs):
            _same(out[off[i]:off[i + 1]], z.compress_chunked(r, zs, 1 << 20, fast=True), (strategy, n, i, len(r)))
    n, stride = 70_000, 4096  # 287 MB: two passes over the scratch buffers (65 535 records to a pass)
    d_in = torch.empty(n * stride, dtype=torch.uint8, device="cuda")
    host = torch.empty(n * stride, dtype=torch.uint8)
    za.corpus_into("records", host.data_ptr(), n * stride, seed=9)
    d_in.copy_(host)
    cap = za.compress_batch_bound(stride, n, st)
    d_out = torch.full((cap,), 0xEE, dtype=torch.uint8, device="cuda")
    d_off = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    total = za.compress_batch_device(d_in.data_ptr(), stride, 0, n, d_out.data_ptr(), cap, d_off.data_ptr(), st)
    off = d_off.cpu().tolist()
    out = d_out[:total].cpu().numpy().tobytes()
    data = host.numpy().tobytes()
    for i in list(range(0, n, 211)) + [65534, 65535, 65536, n - 1]:  # the pass seam is at record 65535
        assert sysz.decompress(out[off[i]:off[i + 1]]) == data[i * stride:(i + 1) * stride], i
    for i in (0, 65534, 65535, n - 1):
        _same(out[off[i]:off[i + 1]], z.compress_chunked(data[i * stride:(i + 1) * stride], zs, 1 << 20, fast=True), i)


def test_one_chunk_of_the_accepted_maximum(za):
    """256 MiB of incompressible bytes as ONE block: more than 2^31 payload bits, the largest the 32-bit tile
    offsets inside a block have to carry."""
    n = 256 << 20
    data = za.corpus("random", n, seed=77)
    got = za.compress(data, za.Strategy.Dynamic, chunk_bytes=n)
    assert len(got) > n  # 8-bit codes at best, plus header and trailer
    d = sysz.decompressobj()
    pos = 0
    for at in range(0, len(got), 8 << 20):
        out = d.decompress(got[at:at + (8 << 20)])
        assert out == data[pos:pos + len(out)]
        pos += len(out)
    assert pos == n and d.eof


def test_blocks_inside_a_very_large_input_equal_the_oracle(za, z):
    """576 MiB of the mixed corpus in one call: three passes over the scratch buffers (256 MiB each) and several
    thousand index segments.  Chunks next to the pass seams, at both ends and in between are compared bit for
    bit with the block the oracle writes for that chunk and its 32 KiB window; the stream as a whole inflates."""
    import torch
    chunk, n_chunks = 1 << 20, 576
    n = chunk * n_chunks
    data = za.corpus("mixed", n, seed=0x5EED0004)
    d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    cap = za.compress_bound(n, za.Strategy.Dynamic, chunk_bytes=chunk)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    ctx = za.default_context()
    out_n = za.compress_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, za.Strategy.Dynamic, chunk_bytes=chunk, ctx=ctx)
    stream = d_out[:out_n].cpu().numpy().tobytes()
    del d_in, d_out
    offs = za.last_block_offsets(ctx=ctx)
    assert len(offs) == n_chunks and offs[0] == 16 and all(a < b for a, b in zip(offs, offs[1:]))
    end_bits = (out_n - 4) * 8  # the last block ends within the last byte before the Adler-32
    for c in (0, 1, 2, 3, 100, 254, 255, 256, 257, 258, 259, 383, 510, 511, 512, 513, 514, 515, 574, 575):
        toks = z.chunk_tokens(data, chunk, c, fast=True)
        want, bits = z.write_block(toks, z.DYNAMIC, c == n_chunks - 1)
        nxt = offs[c + 1] if c + 1 < n_chunks else None
        if nxt is not None:
            assert nxt - offs[c] == bits, f"chunk {c}: block of {nxt - offs[c]} bits, the oracle's has {bits}"
        else:
            assert end_bits - 7 <= offs[c] + bits <= end_bits
        lo, sh = offs[c] // 8, offs[c] % 8
        got = (int.from_bytes(stream[lo:lo + (bits + sh + 7) // 8 + 1], "little") >> sh) & ((1 << bits) - 1)
        assert got == int.from_bytes(want, "little") & ((1 << bits) - 1), f"chunk {c}: block differs from the oracle's"
    d = sysz.decompressobj()
    pos = 0
    for at in range(0, len(stream), 16 << 20):
        out = d.decompress(stream[at:at + (16 << 20)])
        assert out == data[pos:pos + len(out)]
        pos += len(out)
    assert pos == n and d.eof
