"""zg_decompress / zg_decompress_batch: the device inflater against system zlib, the oracle's restatement of
the reference decoder, and the reference's own decoder vectors."""
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


def _inputs(za):
    rng = random.Random(21)
    return [b"", b"a", b"compressed", b"hello hello hello hello", b"a" * 5000, bytes(70_000),
            bytes(rng.randrange(256) for _ in range(3000)), za.corpus("text", 100_000), za.corpus("binary", 60_000),
            za.corpus("mixed", 200_000, seed=3), (b"0123456789abcdef" * 30 + b"x") * 200]


def test_reference_doc_test_stream(za):
    """src/decompress.rs:20-23: a stream written by real zlib."""
    data = bytes([0x78, 0x9C, 0x4B, 0xCE, 0xCF, 0x2D, 0x28, 0x4A, 0x2D, 0x2E, 0x4E, 0x4D, 0x01, 0x00, 0x17, 0x3F, 0x04, 0x36])
    assert za.decompress(data) == b"compressed"


def test_inflates_system_zlib_streams(za):
    """Stored, fixed and dynamic blocks, multi-block streams, every level of system zlib."""
    for data in _inputs(za):
        for level in (0, 1, 6, 9):
            assert za.decompress(sysz.compress(data, level)) == data, (len(data), level)
    big = za.corpus("text", 1_500_000, seed=8)
    c = sysz.compressobj(6)
    stream = b"".join(c.compress(big[i:i + 100_000]) + c.flush(sysz.Z_FULL_FLUSH) for i in range(0, len(big), 100_000))
    stream += c.flush()
    assert za.decompress(stream) == big


def test_round_trips_own_streams(za):
    from oracle import zref as z
    for data in _inputs(za):
        for st in (za.Strategy.Dynamic, za.Strategy.Fixed, za.Strategy.Raw):
            stream = za.compress(data, st, chunk_bytes=65536)
            assert za.decompress(stream) == data
            assert z.decompress(stream) == data  # the restated reference decoder agrees


def test_batch_of_record_streams(za):
    recs = [za.corpus("records", 4096, offset=4096 * i) for i in range(700)] + [b"", b"x", bytes(4096)]
    streams = za.compress_batch(recs)
    assert za.decompress_batch(streams) == recs
    assert za.decompress_batch([]) == []


def test_errors_of_the_reference(za):
    """src/decompress.rs:196-286: the header and checksum errors decompress() returns as Err."""
    good = sysz.compress(b"hello world, hello world", 6)

    def fails(stream, text):
        with pytest.raises(za.ZgError) as e:
            za.decompress(stream)
        assert e.value.code == -8 and text in str(e.value), (e.value.code, str(e.value))

    fails(bytes([0x77, 0x01]) + good[2:], "CM")                      # CM != 8
    fails(bytes([0x88, 0x1C]) + good[2:], "compression info")        # CINFO = 8 (0x881C is a multiple of 31)
    fails(bytes([0x78, 0x02]) + good[2:], "checksum failed")         # FCHECK
    fails(bytes([0x78, 0x20 | (31 - (0x7820 % 31))]) + good[2:], "dictionar")   # FDICT
    fails(bytes([0x78, 0x01, 0x07]), "block type")                   # BFINAL = 1, BTYPE = 3
    fails(good[:-1] + bytes([good[-1] ^ 1]), "Checksum is invalid")
    fails(good[:10], "input ends")
    fails(bytes([0x78, 0x01, 0x4B, 0x04]), "input ends")             # body cut before the end-of-block symbol
    # a match that reaches before the start of the output (fixed block: length 3, distance 1, nothing written yet)
    fails(bytes([0x78, 0x01, 0x03, 0x02, 0x00, 0x00, 0x00, 0x00, 0x01]), "")


def _blocks_round_trip(za, data, chunk, strategy):
    """compress on the device, then inflate all blocks at once from the block offsets the compressor reports"""
    import torch
    n = len(data)
    d_in = torch.frombuffer(bytearray(data) or bytearray(1), dtype=torch.uint8).cuda()
    cap = za.compress_bound(n, strategy, chunk_bytes=chunk)
    d_z = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    ctx = za.default_context()
    zn = za.compress_device(d_in.data_ptr(), n, d_z.data_ptr(), cap, strategy, chunk_bytes=chunk, ctx=ctx)
    offs = za.last_block_offsets(ctx=ctx)
    d_out = torch.zeros(max(1, n), dtype=torch.uint8, device="cuda")
    za.decompress_blocks_device(d_z.data_ptr(), zn, offs, chunk, d_out.data_ptr(), n, ctx=ctx)
    return d_z[:zn].cpu().numpy().tobytes(), offs, d_out[:n].cpu().numpy().tobytes()


@pytest.mark.parametrize("strategy", ["Dynamic", "Fixed"])
def test_all_blocks_of_a_stream_at_once(za, strategy):
    """zg_decompress_blocks_device: every block of a multi-block stream decoded by a wave of its own, matches that
    reach back into the block before resolved afterwards; the result must be the input (and what system zlib makes
    of the same stream, decoding block after block as src/decompress.rs:97-118 does)."""
    st = getattr(za.Strategy, strategy)
    rng = random.Random(31)
    period = bytes(rng.randrange(256) for _ in range(40_000))  # matches 40 000 back: across every seam of 32 KiB chunks
    runs = b"".join(bytes([rng.randrange(4)]) * rng.randint(1, 700) for _ in range(3000))
    cases = [(za.corpus("mixed", (5 << 20) + 12345, seed=41), 1 << 20), (za.corpus("text", 700_001, seed=42), 1 << 16),
             (za.corpus("text", 300_000, seed=43), 32768), (period[:30_000] * 12, 32768), (period[:20_000] * 9, 65536),
             (runs, 32768), (bytes(300_000), 65536), (za.corpus("binary", 3 << 16, seed=44), 1 << 16),
             (za.corpus("random", 200_000, seed=45), 65536), (za.corpus("text", 32768, seed=46), 32768),
             (za.corpus("text", 1000, seed=47), 1 << 20), (b"", 1 << 20), (b"a", 32768)]
    for data, chunk in cases:
        stream, offs, got = _blocks_round_trip(za, data, chunk, st)
        assert len(offs) == max(1, -(-len(data) // chunk))
        assert got == data, (strategy, len(data), chunk)
        assert sysz.decompress(stream) == data


def test_all_blocks_in_several_passes(za, monkeypatch):
    """Streams larger than the scratch buffers take several passes, two of them in flight (the tokens of one are
    expanded while the next is decoded); here with passes of 1 MiB: 1, 2, 3 and 7 passes, blocks of 64 KiB and 1 MiB,
    matches that reach across the seams of blocks and passes."""
    monkeypatch.setenv("ZLIB_AMD_INFLATE_PASS_MIB", "1")
    rng = random.Random(32)
    period = bytes(rng.randrange(256) for _ in range(30_000))
    for data, chunk in ((za.corpus("text", 1 << 20, seed=61), 1 << 16), (za.corpus("mixed", (2 << 20) - 5, seed=62), 1 << 16),
                        (period * 100, 1 << 16), (za.corpus("mixed", (6 << 20) + 4321, seed=63), 1 << 20),
                        (za.corpus("text", (3 << 20) + 1, seed=64), 1 << 20)):
        stream, offs, got = _blocks_round_trip(za, data, chunk, za.Strategy.Dynamic)
        assert got == data, (len(data), chunk)
    # the same call twice in a row (the halves of the buffers and the events are reused)
    stream, offs, got = _blocks_round_trip(za, data, chunk, za.Strategy.Dynamic)
    assert got == data


def test_all_blocks_at_once_refuses_what_does_not_fit(za):
    import torch
    data = za.corpus("text", 200_000, seed=51)
    chunk = 65536
    stream, offs, got = _blocks_round_trip(za, data, chunk, za.Strategy.Dynamic)
    assert got == data
    ctx = za.default_context()
    d_out = torch.zeros(len(data), dtype=torch.uint8, device="cuda")

    def run(s, o, n=len(data), ch=chunk):
        d_z = torch.frombuffer(bytearray(s), dtype=torch.uint8).cuda()
        za.decompress_blocks_device(d_z.data_ptr(), len(s), o, ch, d_out.data_ptr(), n, ctx=ctx)

    run(stream, offs)  # the baseline passes
    assert d_out.cpu().numpy().tobytes() == data
    for bad in ([offs[0]] + [o + 1 for o in offs[1:]], offs[:1] + [offs[1] - 3] + offs[2:]):  # blocks do not begin there
        with pytest.raises(za.ZgError) as e:
            run(stream, bad)
        assert e.value.code == -8
    for at in (len(stream) // 2, 40, len(stream) - 2):  # a flipped bit in a block / in the Adler-32
        broken = bytearray(stream)
        broken[at] ^= 0x10
        with pytest.raises(za.ZgError) as e:
            run(bytes(broken), offs)
        assert e.value.code == -8
    with pytest.raises(za.ZgError) as e:  # header: CM is not 8 (src/decompress.rs:44-46)
        run(b"\x77" + stream[1:], offs)
    assert e.value.code == -8
    for kw in (dict(n=len(data) + chunk), dict(ch=16384), dict(o=offs[:-1])):  # arguments that contradict each other
        with pytest.raises(za.ZgError) as e:
            run(stream, kw.pop("o", offs), **kw)
        assert e.value.code in (-1, -6)
