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
