"""Host-side checks that need no GPU: the C-ABI library loads, exports every
function include/zlib_amd.h declares, refuses to compute without a device, and
its host helpers (corpus generator, Adler combine, compress bound) behave."""
import ctypes as C
import os
import re
import zlib as sysz

import pytest

import zlib_amd
from zlib_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "zlib_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/zlib_amd.h but not exported"
    assert sorted(_lib.SYMBOLS) == names


def _no_device():
    ctx = C.c_void_p()
    return _lib.lib().zg_init(0, C.byref(ctx)) == _lib.ERR_NO_DEVICE


def test_compute_fails_loudly_without_a_device():
    if not _no_device():
        pytest.skip("a HIP device is present")
    with pytest.raises(zlib_amd.ZgError) as e:
        zlib_amd.compress(b"hello hello hello", zlib_amd.Strategy.Dynamic)
    assert e.value.code == _lib.ERR_NO_DEVICE
    out = C.c_uint32()
    assert _lib.lib().zg_adler32(None, b"abc", 3, C.byref(out)) == _lib.ERR_NO_DEVICE


def test_compress_bound_covers_worst_cases():
    for n in (0, 1, 1000, 1 << 20, (1 << 20) + 1, 5 << 20):
        raw = zlib_amd.compress_bound(n, zlib_amd.Strategy.Raw)
        blocks = max(1, -(-n // 16384))
        assert raw >= 2 + n + 5 * blocks + 4
        dyn = zlib_amd.compress_bound(n, zlib_amd.Strategy.Dynamic)
        assert dyn >= n * 15 // 8 + 6
        assert zlib_amd.compress_bound(n, zlib_amd.Strategy.Auto) >= raw - 3
    o = _lib.Opts(1000, 0)  # chunk size must be a multiple of 1024
    assert _lib.lib().zg_compress_bound(10, 1, C.byref(o)) == 0


def test_corpus_is_deterministic_and_offset_addressable():
    for kind in ("text", "random", "binary", "mixed", "records"):
        a = zlib_amd.corpus(kind, 3 << 20, seed=9)
        assert a == zlib_amd.corpus(kind, 3 << 20, seed=9)
        assert a[1_500_000:1_500_000 + 70_000] == zlib_amd.corpus(kind, 70_000, seed=9, offset=1_500_000)
        assert a != zlib_amd.corpus(kind, 3 << 20, seed=10)
    text = zlib_amd.corpus("text", 1 << 20)
    ratio = len(sysz.compress(text, 6)) / len(text)
    assert 0.25 < ratio < 0.5  # text-like
    rnd = zlib_amd.corpus("random", 1 << 20)
    assert len(sysz.compress(rnd, 6)) > len(rnd) * 0.99
    rec = zlib_amd.corpus("records", 4096 * 4096)
    assert any(rec[i * 4096:(i + 1) * 4096] == bytes(4096) for i in range(4096))  # degenerate records exist in the stream


def test_python_mirror_surface():
    c = zlib_amd.LZ77Compressor.new()
    assert (c.window_size, c.min_match_length, c.max_match_length) == (144, 5, 100)  # src/lz77.rs:18-21
    assert zlib_amd.LZ77Compressor.with_window_size((1 << 15) + 1).window_size == 1 << 15  # :90-96
    assert c.set_window_size(1024) is c and c.window_size == 1024
    z = zlib_amd.zlib_compressor()
    assert (z.window_size, z.min_match_length, z.max_match_length) == (32768, 3, 258)
    assert [s.name for s in zlib_amd.Strategy] == ["Auto", "Dynamic", "Fixed", "Raw"]  # src/compress.rs:28-34
    units = [zlib_amd.Literal(65), zlib_amd.Marker(4, 7)]
    assert zlib_amd.unpack_units(zlib_amd.pack_units(units)) == units
    with pytest.raises(ValueError):  # check_lz77, src/huffman.rs:817-835
        zlib_amd.HuffmanTree.from_lz77([], zlib_amd.LZ77Compressor.with_window_size(1024))
