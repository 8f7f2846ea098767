"""The stream / file front end (zg_compress_stream, zg_compress_file): input pulled through a callback segment by
segment, the stream pushed through another.  The bytes must be those zg_compress returns for the whole input
(src/compress.rs:51-86 takes the whole slice), whatever the segment size and however the reader cuts its deliveries."""
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
def test_stream_equals_the_whole_input_call(za, strategy):
    s = getattr(za.Strategy, strategy)
    rng = random.Random(77)
    kw = dict(chunk_bytes=16384, batch_bytes=65536)  # segments of four chunks
    # lengths around the segment seams: a segment that ends the input exactly, one byte more, one byte less
    for n in (0, 1, 255, 300, 1500, 65535, 65536, 65537, 131072, 131073, 200_000, 3 * 65536 + 16385):
        data = za.corpus("mixed", n, seed=5) if n > 4096 else bytes(rng.randrange(97, 101) for _ in range(n))
        want = za.compress(data, s, **kw)
        assert _stream(za, data, s, **kw) == want, (strategy, n)
        assert _stream(za, data, s, reader=_Dribble(data, n), **kw) == want, (strategy, n, "dribble")
        assert sysz.decompress(want) == data


def test_stream_with_default_segments_and_a_size_hint(za):
    data = za.corpus("mixed", 5 * (1 << 20) + 12345, seed=9)
    want = za.compress(data, za.Strategy.Dynamic)
    assert _stream(za, data, za.Strategy.Dynamic, size_hint=len(data)) == want
    assert _stream(za, data, za.Strategy.Dynamic, size_hint=1000) == want  # a hint may be wrong: only buffer sizes follow it
    assert _stream(za, data, za.Strategy.Dynamic, batch_bytes=2 << 20) == want  # three segments, shared seam bytes


def test_chunks_that_are_no_multiple_of_the_stored_block(za):
    data = za.corpus("text", 100_000, seed=3)
    for s in (za.Strategy.Dynamic, za.Strategy.Raw):
        kw = dict(chunk_bytes=3072, batch_bytes=3072)  # the segment becomes 3072 * 16 bytes
        assert _stream(za, data, s, **kw) == za.compress(data, s, **kw)


def test_file_to_file(za, tmp_path):
    data = za.corpus("binary", 3 * (1 << 20) + 17, seed=4)
    src, dst = tmp_path / "in.bin", tmp_path / "out.z"
    src.write_bytes(data)
    n_in, n_out = za.compress_file(str(src), str(dst), za.Strategy.Dynamic, batch_bytes=1 << 20)
    got = dst.read_bytes()
    assert (n_in, n_out) == (len(data), len(got))
    assert got == za.compress(data, za.Strategy.Dynamic)
    with pytest.raises(za.ZgError) as e:
        za.compress_file(str(tmp_path / "missing"), str(dst))
    assert e.value.code == -9


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
