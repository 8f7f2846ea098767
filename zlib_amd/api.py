"""Host-side mirror of the reference crate's public items for the compress
path, on top of the C ABI (same names, argument meaning and error behaviour):

    zlib::compress(&data, &Strategy) -> Vec<u8>        src/compress.rs:51
    zlib::compress::Strategy                            src/compress.rs:28-34
    zlib::adler::adler32                                src/adler.rs:5
    zlib::lz77::{LZ77Unit, LZ77Compressor}              src/lz77.rs:23-38
    zlib::huffman::HuffmanTree::from_lz77 (+ assign, to_canonical, encode)
                                                        src/huffman.rs:238, 619, 668, 714

Failures of `compress` surface as exceptions (the crate panics); nothing here
computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple, Union

from . import _lib
from ._lib import Opts, ShardInfo, ShardPiece, ZgError

ZLIB_WINDOW_SIZE = 1 << 15       # src/huffman.rs:47
ZLIB_MIN_STRING_LENGTH = 3       # src/huffman.rs:48
ZLIB_MAX_STRING_LENGTH = 258     # src/huffman.rs:49
MAX_WINDOW_SIZE = 1 << 15        # src/lz77.rs:20

TOK_MARKER = 0x80000000


class Strategy(enum.IntEnum):
    """src/compress.rs:28-34"""
    Auto = 0
    Dynamic = 1
    Fixed = 2
    Raw = 3


class Context:
    """One HIP stream plus scratch buffers on one device (zg_ctx)."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        rc = _lib.lib().zg_init(device, C.byref(self._h))
        if rc:
            raise ZgError(rc)
        self.device = device

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            _lib.lib().zg_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- measurement -------------------------------------------------------
    def profile(self, on: bool = True, only: Optional[str] = None):
        """Phase timers on or off; `only`: the name of the one kernel to time (as profile_get reports it)."""
        mode = int(bool(on))
        if on and only is not None:
            mode = 2 + list(self.profile_get()).index(only)
        _check(_lib.lib().zg_profile_enable(self._h, mode), self)

    def profile_reset(self):
        _check(_lib.lib().zg_profile_reset(self._h), self)

    def profile_get(self):
        """{kernel name: (milliseconds, launches)} accumulated since the last reset."""
        L = _lib.lib()
        out = {}
        for i in range(L.zg_profile_count(self._h)):
            name, ms, n = C.c_char_p(), C.c_double(), C.c_uint64()
            _check(L.zg_profile_get(self._h, i, C.byref(name), C.byref(ms), C.byref(n)), self)
            out[name.value.decode()] = (ms.value, n.value)
        return out


_default: Optional[Context] = None


def default_context() -> Context:
    global _default
    if _default is None:
        _default = Context(0)
    return _default


def _check(rc: int, ctx: Optional[Context] = None):
    if rc:
        detail = ""
        if rc in (_lib.ERR_HIP, _lib.ERR_DATA, _lib.ERR_CAPACITY, _lib.ERR_IO) and ctx is not None:
            detail = _lib.lib().zg_last_error(ctx.handle).decode()
        raise ZgError(rc, detail)


def _opts(chunk_bytes: int = 0, batch_bytes: int = 0):
    return Opts(chunk_bytes, batch_bytes)


def compress(data: bytes, strategy: Strategy = Strategy.Auto, *, chunk_bytes: int = 0, batch_bytes: int = 0,
             ctx: Optional[Context] = None) -> bytes:
    """zlib::compress(&data, &strategy)."""
    ctx = ctx or default_context()
    data = bytes(data)
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    o = _opts(chunk_bytes, batch_bytes)
    _check(_lib.lib().zg_compress(ctx.handle, data, len(data), int(strategy), C.byref(o), C.byref(out), C.byref(n)), ctx)
    try:
        return C.string_at(out, n.value)
    finally:
        _lib.lib().zg_free(out)


def compress_bound(n: int, strategy: Strategy = Strategy.Dynamic, *, chunk_bytes: int = 0) -> int:
    o = _opts(chunk_bytes)
    return _lib.lib().zg_compress_bound(n, int(strategy), C.byref(o))


def compress_device(d_in: int, n: int, d_out: int, out_cap: int, strategy: Strategy = Strategy.Dynamic, *,
                    chunk_bytes: int = 0, batch_bytes: int = 0, ctx: Optional[Context] = None) -> int:
    """Device pointers in and out (e.g. torch tensors' data_ptr()); returns the stream length."""
    ctx = ctx or default_context()
    n_out = C.c_size_t()
    o = _opts(chunk_bytes, batch_bytes)
    _check(_lib.lib().zg_compress_device(ctx.handle, d_in, n, int(strategy), C.byref(o), d_out, out_cap, C.byref(n_out)), ctx)
    return n_out.value


def compress_multi(data: bytes, devices: Sequence[int], strategy: Strategy = Strategy.Dynamic, *, chunk_bytes: int = 0,
                   batch_bytes: int = 0) -> bytes:
    """zlib::compress(&data, &strategy) on several GPUs of this process: one run of chunks per entry of `devices`."""
    data = bytes(data)
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    o = _opts(chunk_bytes, batch_bytes)
    devs = (C.c_int * max(1, len(devices)))(*devices)
    rc = _lib.lib().zg_compress_multi(devs, len(devices), data, len(data), int(strategy), C.byref(o), C.byref(out), C.byref(n))
    if rc:
        raise ZgError(rc)
    try:
        return C.string_at(out, n.value)
    finally:
        _lib.lib().zg_free(out)


def compress_stream(reader, writer, strategy: Strategy = Strategy.Dynamic, *, chunk_bytes: int = 0,
                    batch_bytes: int = 0, size_hint: int = 0, ctx: Optional[Context] = None) -> Tuple[int, int]:
    """zlib::compress for input that is not in memory as a whole: `reader.readinto(buffer)` (or `.read(n)`)
    delivers the input, `writer.write(bytes)` receives the stream, a segment of `batch_bytes` at a time while the
    neighbouring segments are read, uploaded and downloaded.  Returns (input bytes, stream bytes); the bytes
    written are those compress() returns for the whole input.  `ctx` is locked for the whole call: a reader or
    writer that calls this module on the same context would wait for ever (give it a Context of its own)."""
    ctx = ctx or default_context()
    failure: List[BaseException] = []

    def _read(_user, buf, cap):
        try:
            if hasattr(reader, "readinto"):
                view = (C.c_uint8 * cap).from_address(C.addressof(buf.contents))
                got = reader.readinto(memoryview(view).cast("B"))
                return int(got or 0)
            chunk = reader.read(cap)
            C.memmove(buf, chunk, len(chunk))
            return len(chunk)
        except BaseException as e:  # an exception must not cross the C frames: report a failed read
            failure.append(e)
            return C.c_size_t(-1).value

    def _write(_user, buf, n):
        try:
            writer.write(C.string_at(buf, n))
            return 0
        except BaseException as e:
            failure.append(e)
            return -1

    rfn, wfn = _lib.READ_FN(_read), _lib.WRITE_FN(_write)
    o = _opts(chunk_bytes, batch_bytes)
    n_in, n_out = C.c_uint64(), C.c_uint64()
    rc = _lib.lib().zg_compress_stream(ctx.handle, rfn, None, wfn, None, int(strategy), C.byref(o), size_hint,
                                       C.byref(n_in), C.byref(n_out))
    if failure:
        raise failure[0]
    _check(rc, ctx)
    return n_in.value, n_out.value


def compress_file(in_path: str, out_path: str, strategy: Strategy = Strategy.Dynamic, *, chunk_bytes: int = 0,
                  batch_bytes: int = 0, ctx: Optional[Context] = None) -> Tuple[int, int]:
    """compress_stream between two paths (read and written by the library itself)."""
    ctx = ctx or default_context()
    o = _opts(chunk_bytes, batch_bytes)
    n_in, n_out = C.c_uint64(), C.c_uint64()
    import os
    _check(_lib.lib().zg_compress_file(ctx.handle, os.fsencode(in_path), os.fsencode(out_path), int(strategy),
                                       C.byref(o), C.byref(n_in), C.byref(n_out)), ctx)
    return n_in.value, n_out.value


def last_block_offsets(*, ctx: Optional[Context] = None) -> List[int]:
    """Bit offsets of the blocks written by the last compress / compress_device / shard_begin call on ctx."""
    ctx = ctx or default_context()
    n = _lib.lib().zg_last_block_offsets(ctx.handle, None, 0)
    buf = (C.c_uint64 * max(1, n))()
    _lib.lib().zg_last_block_offsets(ctx.handle, buf, n)
    return [int(buf[i]) for i in range(n)]


def compress_batch(records: Sequence[bytes], strategy: Strategy = Strategy.Dynamic, *,
                   ctx: Optional[Context] = None) -> List[bytes]:
    """[zlib::compress(r, &strategy) for r in records], all records in flight on the GPU at once."""
    ctx = ctx or default_context()
    blob = b"".join(bytes(r) for r in records)
    offs = (C.c_size_t * (len(records) + 1))()
    pos = 0
    for i, r in enumerate(records):
        offs[i] = pos
        pos += len(r)
    offs[len(records)] = pos
    out, out_offs = C.POINTER(C.c_uint8)(), C.POINTER(C.c_size_t)()
    _check(_lib.lib().zg_compress_batch(ctx.handle, blob, offs, len(records), int(strategy), C.byref(out),
                                        C.byref(out_offs)), ctx)
    try:
        total = out_offs[len(records)]
        whole = C.string_at(out, total)
        return [whole[out_offs[i]:out_offs[i + 1]] for i in range(len(records))]
    finally:
        _lib.lib().zg_free(out)
        _lib.lib().zg_free(out_offs)


def decompress(data: bytes, *, ctx: Optional[Context] = None) -> bytes:
    """zlib::decompress(&data): Ok(bytes) or ZgError (ZG_ERR_DATA, the reference's Err text as the message)."""
    ctx = ctx or default_context()
    data = bytes(data)
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    _check(_lib.lib().zg_decompress(ctx.handle, data, len(data), C.byref(out), C.byref(n)), ctx)
    try:
        return C.string_at(out, n.value)
    finally:
        _lib.lib().zg_free(out)


def decompress_batch(streams: Sequence[bytes], *, ctx: Optional[Context] = None) -> List[bytes]:
    """[zlib::decompress(s) for s in streams], all streams in flight on the GPU at once."""
    ctx = ctx or default_context()
    blob = b"".join(bytes(s) for s in streams)
    offs = (C.c_size_t * (len(streams) + 1))()
    pos = 0
    for i, s in enumerate(streams):
        offs[i] = pos
        pos += len(s)
    offs[len(streams)] = pos
    out, out_offs = C.POINTER(C.c_uint8)(), C.POINTER(C.c_size_t)()
    _check(_lib.lib().zg_decompress_batch(ctx.handle, blob, offs, len(streams), C.byref(out), C.byref(out_offs)), ctx)
    try:
        whole = C.string_at(out, out_offs[len(streams)])
        return [whole[out_offs[i]:out_offs[i + 1]] for i in range(len(streams))]
    finally:
        _lib.lib().zg_free(out)
        _lib.lib().zg_free(out_offs)


def decompress_blocks_device(d_stream: int, stream_bytes: int, bit_offsets: Sequence[int], chunk_bytes: int, d_out: int,
                             n: int, *, ctx: Optional[Context] = None) -> None:
    """zlib::decompress of one large stream of this library with every block decoded at once: `bit_offsets` is what
    last_block_offsets() returned after compressing, `n` the length of the uncompressed data; device pointers."""
    ctx = ctx or default_context()
    offs = (C.c_uint64 * max(1, len(bit_offsets)))(*bit_offsets)
    _check(_lib.lib().zg_decompress_blocks_device(ctx.handle, d_stream, stream_bytes, offs, len(bit_offsets), chunk_bytes,
                                                  d_out, n), ctx)


def compress_batch_bound(stride: int, n_records: int, strategy: Strategy = Strategy.Dynamic) -> int:
    return _lib.lib().zg_compress_batch_bound(stride, n_records, int(strategy))


def compress_batch_device(d_in: int, stride: int, d_len: int, n_records: int, d_out: int, out_cap: int,
                          d_out_offsets: int, strategy: Strategy = Strategy.Dynamic, *,
                          ctx: Optional[Context] = None) -> int:
    """Device pointers: records of d_len[i] (uint32; 0 pointer = all `stride`) bytes in slots of `stride` bytes;
    d_out_offsets receives n_records + 1 uint64 offsets.  Returns the total number of output bytes."""
    ctx = ctx or default_context()
    n_out = C.c_size_t()
    _check(_lib.lib().zg_compress_batch_device(ctx.handle, d_in, stride, d_len, n_records, int(strategy), d_out,
                                               out_cap, d_out_offsets, C.byref(n_out)), ctx)
    return n_out.value


def shard_begin(d_in: int, halo: int, n: int, strategy: Strategy, last_rank: bool, *, chunk_bytes: int = 0,
                ctx: Optional[Context] = None) -> ShardInfo:
    ctx = ctx or default_context()
    info = ShardInfo()
    o = _opts(chunk_bytes)
    _check(_lib.lib().zg_shard_begin(ctx.handle, d_in, halo, n, int(strategy), C.byref(o), int(last_rank), C.byref(info)), ctx)
    return info


def shard_finish(base_bit: int, first_rank: bool, last_rank: bool, adler: int, d_out: int, out_cap: int, *,
                 ctx: Optional[Context] = None) -> int:
    ctx = ctx or default_context()
    n_out = C.c_size_t()
    _check(_lib.lib().zg_shard_finish(ctx.handle, base_bit, int(first_rank), int(last_rank), adler, d_out, out_cap,
                                      C.byref(n_out)), ctx)
    return n_out.value


def shard_open(d_in: int, halo: int, n: int, strategy: Strategy, last_rank: bool, d_packed: int, packed_cap: int, *,
               chunk_bytes: int = 0, batch_bytes: int = 0, ctx: Optional[Context] = None) -> None:
    """zg_shard_open: starts a rank's run; its blocks are packed into the caller's device buffer pass by pass."""
    ctx = ctx or default_context()
    o = _opts(chunk_bytes, batch_bytes)
    _check(_lib.lib().zg_shard_open(ctx.handle, d_in, halo, n, int(strategy), C.byref(o), int(last_rank), d_packed,
                                    packed_cap), ctx)


def shard_next(*, ctx: Optional[Context] = None):
    """zg_shard_next: (byte_offset, n_bytes, bits_so_far, last, info or None) of the oldest unreported pass."""
    ctx = ctx or default_context()
    piece, info = ShardPiece(), ShardInfo()
    _check(_lib.lib().zg_shard_next(ctx.handle, C.byref(piece), C.byref(info)), ctx)
    return piece.byte_offset, piece.n_bytes, piece.bits_so_far, bool(piece.last), (info if piece.last else None)


def shard_merge(d_src: int, n_bits: int, d_dst: int, dst_cap: int, base_bit: int, first_run: bool, *,
                ctx: Optional[Context] = None) -> None:
    ctx = ctx or default_context()
    _check(_lib.lib().zg_shard_merge(ctx.handle, d_src, n_bits, d_dst, dst_cap, base_bit, int(first_run)), ctx)


def shard_seal(d_dst: int, dst_cap: int, end_bit: int, adler: int, *, ctx: Optional[Context] = None) -> int:
    ctx = ctx or default_context()
    n_out = C.c_size_t()
    _check(_lib.lib().zg_shard_seal(ctx.handle, d_dst, dst_cap, end_bit, adler, C.byref(n_out)), ctx)
    return n_out.value


def adler32_combine(adler1: int, adler2: int, len2: int) -> int:
    return _lib.lib().zg_adler32_combine(adler1, adler2, len2)


def adler32(data: bytes, *, ctx: Optional[Context] = None) -> int:
    """zlib::adler::adler32(&data)."""
    ctx = ctx or default_context()
    data = bytes(data)
    out = C.c_uint32()
    _check(_lib.lib().zg_adler32(ctx.handle, data, len(data), C.byref(out)), ctx)
    return out.value


def adler32_device(d_in: int, n: int, *, ctx: Optional[Context] = None) -> int:
    """Adler-32 of n bytes of device memory at d_in."""
    ctx = ctx or default_context()
    out = C.c_uint32()
    _check(_lib.lib().zg_adler32_device(ctx.handle, d_in, n, C.byref(out)), ctx)
    return out.value


def adler32_device_noalloc(d_in: int, n: int, d_out: int, *, ctx: Context) -> None:
    """zg_adler32_device_noalloc: the checksum goes to the uint32 at DEVICE address d_out; allocates nothing."""
    _check(_lib.lib().zg_adler32_device_noalloc(ctx.handle, d_in, n, d_out), ctx)


@dataclass(frozen=True, order=True)
class Literal:
    """LZ77Unit::Literal(u8)"""
    byte: int


@dataclass(frozen=True, order=True)
class Marker:
    """LZ77Unit::Marker(length, distance)"""
    length: int
    distance: int


LZ77Unit = Union[Literal, Marker]


def unpack_units(tokens: Sequence[int]) -> List[LZ77Unit]:
    return [Marker((t >> 16) & 0x1FF, t & 0xFFFF) if t & TOK_MARKER else Literal(t & 0xFF) for t in tokens]


def pack_units(units: Sequence[LZ77Unit]) -> List[int]:
    return [TOK_MARKER | (u.length << 16) | u.distance if isinstance(u, Marker) else u.byte for u in units]


class LZ77Compressor:
    """src/lz77.rs:31-207: public fields window_size / min_match_length /
    max_match_length with the crate's defaults (144 / 5 / 100)."""

    def __init__(self):
        self.window_size = 144
        self.min_match_length = 5
        self.max_match_length = 100

    @classmethod
    def new(cls):
        return cls()

    @classmethod
    def with_window_size(cls, window_size: int):
        c = cls()
        c.window_size = min(window_size, MAX_WINDOW_SIZE)
        return c

    def set_window_size(self, newsize: int):
        self.window_size = min(newsize, MAX_WINDOW_SIZE)
        return self

    def compress_packed(self, data: bytes, *, start: int = 0, ctx: Optional[Context] = None) -> List[int]:
        ctx = ctx or default_context()
        data = bytes(data)
        toks, n = C.POINTER(C.c_uint32)(), C.c_size_t()
        _check(_lib.lib().zg_lz77(ctx.handle, data, len(data), start, self.window_size, self.min_match_length,
                                  self.max_match_length, C.byref(toks), C.byref(n)), ctx)
        try:
            return [toks[i] for i in range(n.value)]
        finally:
            _lib.lib().zg_free(toks)

    def compress(self, data: bytes, *, ctx: Optional[Context] = None) -> List[LZ77Unit]:
        """LZ77Compressor::compress(&self, &data) -> Vec<LZ77Unit>"""
        return unpack_units(self.compress_packed(data, ctx=ctx))


def zlib_compressor() -> LZ77Compressor:
    """get_zlib_compressor, src/compress.rs:451-456"""
    c = LZ77Compressor.with_window_size(ZLIB_WINDOW_SIZE)
    c.min_match_length = ZLIB_MIN_STRING_LENGTH
    c.max_match_length = ZLIB_MAX_STRING_LENGTH
    return c


class HuffmanTree:
    """The encoding side of zlib::huffman::HuffmanTree: a canonical code table
    (symbol -> (LSB-first code, length)) as `assign` + `to_canonical` leave it."""

    def __init__(self, table=None):
        self._map = dict(table or {})

    def encode(self, symbol: int) -> Optional[Tuple[int, int]]:
        return self._map.get(int(symbol))

    def encodings(self):
        return dict(self._map)

    def n_codes(self) -> int:
        return len(self._map)

    def max_code_len(self) -> int:
        return max((l for _, l in self._map.values()), default=0)

    @staticmethod
    def from_lz77(units: Sequence, lz77: Optional[LZ77Compressor] = None, *, ctx: Optional[Context] = None):
        """HuffmanTree::from_lz77(&units, &lz77) followed by assign + to_canonical on both trees
        (create_dynamic_trees, src/compress.rs:296-314).  `lz77` must carry the zlib parameters
        (check_lz77, src/huffman.rs:817-835)."""
        if lz77 is not None:
            if lz77.window_size != ZLIB_WINDOW_SIZE:
                raise ValueError(f"Incompatible LZ77 window size, required {ZLIB_WINDOW_SIZE}, found {lz77.window_size}")
            if lz77.min_match_length != ZLIB_MIN_STRING_LENGTH:
                raise ValueError(f"Incompatible LZ77 min string length, required 3, found {lz77.min_match_length}")
            if lz77.max_match_length != ZLIB_MAX_STRING_LENGTH:
                raise ValueError(f"Incompatible LZ77 max string length, required 258, found {lz77.max_match_length}")
        ctx = ctx or default_context()
        toks = list(units)
        if toks and not isinstance(toks[0], int):
            toks = pack_units(toks)
        arr = (C.c_uint32 * max(1, len(toks)))(*toks)
        ll_len, d_len = (C.c_uint8 * 286)(), (C.c_uint8 * 30)()
        ll_code, d_code = (C.c_uint16 * 286)(), (C.c_uint16 * 30)()
        _check(_lib.lib().zg_huffman_from_tokens(ctx.handle, arr, len(toks), ll_len, ll_code, d_len, d_code), ctx)
        lt = HuffmanTree({s: (ll_code[s], ll_len[s]) for s in range(286) if ll_len[s]})
        dt = HuffmanTree({s: (d_code[s], d_len[s]) for s in range(30) if d_len[s]})
        return lt, dt


CORPUS_KINDS = {"text": 0, "random": 1, "binary": 2, "mixed": 3, "records": 4}


def corpus(kind: str, n: int, seed: int = 0x5EED0001, offset: int = 0) -> bytes:
    """Synthetic corpus bytes [offset, offset + n) (host-side generator)."""
    buf = C.create_string_buffer(max(1, n))
    rc = _lib.lib().zg_corpus_fill(CORPUS_KINDS[kind], seed, offset, C.cast(buf, C.c_void_p), n)
    if rc:
        raise ZgError(rc)
    return buf.raw[:n]


def corpus_into(kind: str, ptr: int, n: int, seed: int = 0x5EED0001, offset: int = 0):
    """Fills host memory at `ptr` (e.g. a pinned torch tensor's data_ptr())."""
    rc = _lib.lib().zg_corpus_fill(CORPUS_KINDS[kind], seed, offset, ptr, n)
    if rc:
        raise ZgError(rc)
