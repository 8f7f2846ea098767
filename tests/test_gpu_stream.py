"""The stream / file front end (zg_compress_stream, zg_compress_file): input pulled through a callback segment by
segment, the stream pushed through another.  The bytes must be those of compress on the whole input
(src/compress.rs:51-86 takes the whole slice) -- checked against the ORACLE's stream for that input, and against
zg_compress -- whatever the segment size and however the reader cuts its deliveries."""
import io
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


class _Dribble(io.RawIOBase):
    """A reader that delivers its bytes in pieces of irregular size, like a pipe."""

    def __init__(self, data, seed):
        self.data, self.pos, self.rng = data, 0, random.Random(seed)

    def readable(self):
        return True

    def readinto(self, b):
        n = min(len(b), len(self.data) - self.pos, self.rng.choice((1, 7, 4096, 65536, 1 << 20)))
        b[:n] = self.data[self.pos:self.pos + n]
        self.pos += n
        return n


def _stream(za, data, strategy, reader=None, **kw):
    sink = io.BytesIO()
    n_in, n_out = za.compress_stream(reader or io.BytesIO(data), sink, strategy, **kw)
    got = sink.getvalue()
    assert (n_in, n_out) == (len(data), len(got))
    return got


@pytest.mark.parametrize("strategy", ["Dynamic", "Fixed", "Raw", "Auto"])
def test_stream_equals_the_whole_input_call(za, z, strategy):
    s, zs = getattr(za.Strategy, strategy), getattr(z, strategy.upper())
    rng = random.Random(77)
    kw = dict(chunk_bytes=16384, batch_bytes=65536)  # segments of four chunks
    # lengths around the segment seams: a segment that ends the input exactly, one byte more, one byte less
    for n in (0, 1, 255, 300, 1500, 65535, 65536, 65537, 131072, 131073, 200_000, 3 * 65536 + 16385):
        data = za.corpus("mixed", n, seed=5) if n > 4096 else bytes(rng.randrange(97, 101) for _ in range(n))
        want = z.compress_chunked(data, zs, 16384, fast=True)  # the oracle's stream for the whole input
        assert za.compress(data, s, **kw) == want, (strategy, n, "whole-input call")
        assert _stream(za, data, s, **kw) == want, (strategy, n)
        assert _stream(za, data, s, reader=_Dribble(data, n), **kw) == want, (strategy, n, "dribble")
        assert sysz.decompress(want) == data


def test_stream_with_default_segments_and_a_size_hint(za, z):
    data = za.corpus("mixed", 5 * (1 << 20) + 12345, seed=9)
    want = z.compress_chunked(data, z.DYNAMIC, 1 << 20, fast=True)
    assert za.compress(data, za.Strategy.Dynamic) == want
    assert _stream(za, data, za.Strategy.Dynamic, size_hint=len(data)) == want
    assert _stream(za, data, za.Strategy.Dynamic, size_hint=1000) == want  # a hint may be wrong: only buffer sizes follow it
    assert _stream(za, data, za.Strategy.Dynamic, batch_bytes=2 << 20) == want  # three segments, shared seam bytes


def test_chunks_that_are_no_multiple_of_the_stored_block(za, z):
    data = za.corpus("text", 100_000, seed=3)
    for s, zs in ((za.Strategy.Dynamic, z.DYNAMIC), (za.Strategy.Raw, z.RAW)):
        kw = dict(chunk_bytes=3072, batch_bytes=3072)  # the segment becomes 3072 * 16 bytes
        want = z.compress_chunked(data, zs, 3072, fast=True)
        assert _stream(za, data, s, **kw) == want
        assert za.compress(data, s, **kw) == want


@pytest.mark.parametrize("strategy", ["Dynamic", "Fixed", "Raw"])
def test_segments_are_never_shorter_than_the_window(za, z, strategy):
    """Chunk and batch sizes that would give segments of 16 KiB: every chunk must still be matched against the whole
    32 KiB before it, as in a call on the whole input (a segment only keeps the halo of the ONE segment before it, so
    segments are rounded up to the window).  Also a size hint below the window on longer input."""
    s, zs = getattr(za.Strategy, strategy), getattr(z, strategy.upper())
    # a period of 20 000 bytes: the only matches lie 20 000 back -- more than 16 KiB, less than the window
    block = za.corpus("random", 20_000, seed=31)
    data = block * 5 + za.corpus("text", 30_000, seed=32)
    for chunk in (16384, 1024):
        want = z.compress_chunked(data, zs, chunk, fast=True)
        assert za.compress(data, s, chunk_bytes=chunk) == want, (strategy, chunk, "whole-input call")
        assert _stream(za, data, s, chunk_bytes=chunk, batch_bytes=chunk) == want, (strategy, chunk)
        assert _stream(za, data, s, reader=_Dribble(data, chunk), chunk_bytes=chunk, batch_bytes=chunk) == want
    want = z.compress_chunked(data, zs, 16384, fast=True)
    assert _stream(za, data, s, chunk_bytes=16384, size_hint=5000) == want
    assert sysz.decompress(want) == data


def test_file_to_file(za, tmp_path):
    data = za.corpus("binary", 3 * (1 << 20) + 17, seed=4)
    src, dst = tmp_path / "in.bin", tmp_path / "out.z"
    src.write_bytes(data)
    n_in, n_out = za.compress_file(str(src), str(dst), za.Strategy.Dynamic, batch_bytes=1 << 20)
    got = dst.read_bytes()
    assert (n_in, n_out) == (len(data), len(got))
    assert got == za.compress(data, za.Strategy.Dynamic)
    from oracle import zref
    assert got == zref.compress_chunked(data, zref.DYNAMIC, 1 << 20, fast=True)
    with pytest.raises(za.ZgError) as e:
        za.compress_file(str(tmp_path / "missing"), str(dst))
    assert e.value.code == -9
    # input and output must not be the same file (the output is truncated before the first read); the input survives
    with pytest.raises(za.ZgError) as e:
        za.compress_file(str(src), str(src))
    assert e.value.code == -1 and src.read_bytes() == data
    link = tmp_path / "alias.bin"
    link.symlink_to(src)
    with pytest.raises(za.ZgError):
        za.compress_file(str(src), str(link))
    assert src.read_bytes() == data
    # a call that fails leaves no partial output behind
    bad = tmp_path / "dir_as_input"
    bad.mkdir()
    out2 = tmp_path / "never.z"
    with pytest.raises(za.ZgError):
        za.compress_file(str(bad), str(out2))
    assert not out2.exists()


def test_failures_of_the_callbacks_end_the_call(za):
    data = za.corpus("text", 300_000, seed=1)

    class Broken(io.RawIOBase):
        def __init__(self):
            self.n = 0

        def readinto(self, b):
            self.n += 1
            if self.n > 2:
                raise OSError("the pipe broke")
            b[:1000] = data[:1000]
            return 1000

    with pytest.raises(OSError, match="the pipe broke"):
        za.compress_stream(Broken(), io.BytesIO(), za.Strategy.Dynamic, chunk_bytes=16384, batch_bytes=65536)

    class Full:
        def write(self, b):
            raise OSError("no space left")

    with pytest.raises(OSError, match="no space left"):
        za.compress_stream(io.BytesIO(data), Full(), za.Strategy.Dynamic, chunk_bytes=16384, batch_bytes=65536)
    # the context is usable afterwards
    assert sysz.decompress(za.compress(data, za.Strategy.Dynamic)) == data
