"""ctypes binding of the CPU oracle (oracle/libzref.so).

TEST INFRASTRUCTURE ONLY: used by tests/, the smoke check and the
cpu_baseline leg of bench.py.  The product package never imports this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libzref.so")

AUTO, DYNAMIC, FIXED, RAW = 0, 1, 2, 3
TIE_FIFO, TIE_HEAP = 0, 1
OK = 0
ERR_ARG, ERR_DECODE, ERR_MALFORMED = -1, -2, -3
PANIC_LL_TOO_LONG, PANIC_D_TOO_LONG, PANIC_CL_TOO_LONG = -101, -102, -103
PANIC_NO_ENCODING, PANIC_HCODES, PANIC_RAW_DEFECT, PANIC_ZERO_LEN_CODE = -104, -105, -106, -107

TOK_MARKER = 0x80000000


def build(force: bool = False) -> str:
    """Compile the oracle with the recipe in oracle/Makefile."""
    src = [os.path.join(_HERE, f) for f in ("zref.cpp", "zref.h")]
    stale = force or not os.path.exists(_SO) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in src if os.path.exists(s))
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "libzref.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _declare(_lib)
    return _lib


_sz, _u8p, _u32p, _szp, _vp = C.c_size_t, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_size_t), C.c_void_p
_lp = C.POINTER(C.c_long)


def _declare(L):
    def f(name, res, *args):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = list(args)

    f("zo_free", None, _vp)
    f("zo_bw_new", _vp)
    f("zo_bw_free", None, _vp)
    f("zo_bw_write_bit", None, _vp, C.c_uint)
    f("zo_bw_write_bits", None, _vp, _sz, _sz)
    f("zo_bw_write_byte", None, _vp, C.c_uint)
    f("zo_bw_write_bytes", None, _vp, C.c_char_p, _sz)
    f("zo_bw_flush_byte", None, _vp)
    f("zo_bw_finish", C.c_int, _vp, C.POINTER(_u8p), _szp)
    f("zo_br_new", _vp, C.c_char_p, _sz)
    f("zo_br_free", None, _vp)
    f("zo_br_read_byte", C.c_int, _vp)
    f("zo_br_read_bit", C.c_int, _vp)
    f("zo_br_read_bits", C.c_longlong, _vp, _sz)
    f("zo_br_read_bytes", C.c_longlong, _vp, _sz)
    f("zo_code_to_bytes", C.c_int, _sz, _sz, C.POINTER(_u8p), _szp)
    f("zo_adler32", C.c_uint32, C.c_char_p, _sz)
    for t in ("zo_code_length_codes_order", "zo_length_extra_bits", "zo_length_base",
              "zo_distance_extra_bits", "zo_distance_base"):
        f(t, _szp)
    f("zo_get_length_code", _sz, _sz)
    f("zo_get_distance_code", _sz, _sz)
    f("zo_tree_new", _vp)
    f("zo_tree_free", None, _vp)
    f("zo_tree_insert", None, _vp, _sz, _sz, C.c_uint32)
    f("zo_tree_assign", None, _vp)
    f("zo_tree_encode", C.c_int, _vp, C.c_uint32, _szp, _szp)
    f("zo_tree_n_codes", _sz, _vp)
    f("zo_tree_max_code_len", _sz, _vp)
    f("zo_tree_to_canonical", _vp, _vp)
    f("zo_tree_from_bitlen_list", _vp, _szp, _sz, _u32p, _sz)
    f("zo_tree_from_data", _vp, C.c_char_p, _sz, C.c_int, C.c_uint64)
    f("zo_tree_from_freq", _vp, _u32p, _szp, _sz, C.c_int, C.c_uint64)
    f("zo_tree_get_zlib_fixed", None, C.POINTER(_vp), C.POINTER(_vp))
    f("zo_tree_from_lz77", C.c_int, _u32p, _sz, C.c_int, C.c_uint64, C.POINTER(_vp), C.POINTER(_vp))
    f("zo_tree_decode", C.c_long, _vp, _vp)
    f("zo_tree_decode_trees", C.c_int, _vp, C.POINTER(_vp), C.POINTER(_vp))
    f("zo_tree_encodings", _sz, _vp, _u32p, _szp, _szp, _sz)
    for n in ("zo_lz77_compress", "zo_lz77_compress_literal"):
        f(n, C.c_int, C.c_char_p, _sz, _sz, _sz, _sz, _sz, C.POINTER(_u32p), _szp)
    f("zo_lz77_fast", C.c_int, C.c_char_p, _sz, _sz, _sz, _sz, C.POINTER(_u32p), _szp)
    f("zo_literal_writer", C.c_int, _vp, _vp, C.c_uint32)
    f("zo_length_writer", C.c_int, _vp, _vp, _sz)
    f("zo_distance_writer", C.c_int, _vp, _vp, _sz)
    f("zo_write_compressed_data", C.c_int, _vp, _u32p, _sz, _vp, _vp)
    f("zo_write_dynamic_header", C.c_int, _vp, _vp, _vp, C.c_int)
    f("zo_get_codelengths", C.c_int, _vp, _vp, C.POINTER(_szp), _szp, C.POINTER(_szp), _szp)
    f("zo_zlib_rle", C.c_int, _szp, _sz, _szp, _sz, C.POINTER(_szp), C.POINTER(_lp), _szp)
    f("zo_run_length_encode", C.c_int, _szp, _sz, C.POINTER(_szp), C.POINTER(_szp), _szp)
    f("zo_rle_to_zlib_codes", C.c_int, _szp, _szp, _sz, C.POINTER(_szp), C.POINTER(_lp), _szp)
    f("zo_zlib_rle_encode_zero", C.c_int, _sz, C.POINTER(_szp), C.POINTER(_lp), _szp)
    f("zo_zlib_rle_encode_nonzero", C.c_int, _sz, _sz, C.POINTER(_szp), C.POINTER(_lp), _szp)
    f("zo_get_code_tree", C.c_int, _szp, _sz, C.c_int, C.POINTER(_vp))
    f("zo_get_hcodes", C.c_int, _vp, C.POINTER(_szp), _szp)
    f("zo_compress_reference", C.c_int, C.c_char_p, _sz, C.c_int, C.POINTER(_u8p), _szp)
    f("zo_compress_chunked", C.c_int, C.c_char_p, _sz, C.c_int, _sz, C.c_int, C.POINTER(_u8p), _szp)
    f("zo_chunk_tokens", C.c_int, C.c_char_p, _sz, _sz, _sz, C.c_int, C.POINTER(_u32p), _szp)
    f("zo_chunk_tables", C.c_int, _u32p, _sz, _u8p, C.POINTER(C.c_uint16), _u8p, C.POINTER(C.c_uint16))
    f("zo_block_bits", C.c_int, _u32p, _sz, C.c_int, C.POINTER(C.c_uint64))
    f("zo_write_block", C.c_int, _u32p, _sz, C.c_int, C.c_int, C.POINTER(_u8p), _szp, C.POINTER(C.c_uint64))
    f("zo_auto_choice", C.c_int, C.c_char_p, _sz)
    f("zo_decompress", C.c_int, C.c_char_p, _sz, C.POINTER(_u8p), _szp, C.c_char_p, _sz)
    f("zo_bench_slices", C.c_int, C.c_char_p, _szp, _szp, _szp, _sz, C.c_int, C.POINTER(C.c_uint64))


class OracleError(Exception):
    def __init__(self, code, msg=""):
        super().__init__(f"oracle status {code} {msg}")
        self.code = code
        self.msg = msg


def _take_bytes(p, n) -> bytes:
    out = C.string_at(p, n.value) if n.value else b""
    lib().zo_free(p)
    return out


def _take_list(p, n, conv=int):
    out = [conv(p[i]) for i in range(n)]
    lib().zo_free(p)
    return out


def _sz_array(v):
    return (C.c_size_t * max(1, len(v)))(*v)


def _tok_array(tokens):
    return (C.c_uint32 * max(1, len(tokens)))(*tokens)


# ---- token helpers -------------------------------------------------------
def literal(b: int) -> int:
    return b & 0xFF


def marker(length: int, dist: int) -> int:
    return TOK_MARKER | (length << 16) | dist


def tok_str(t: int) -> str:
    return f"M({(t >> 16) & 0x1FF},{t & 0xFFFF})" if t & TOK_MARKER else f"L({t & 0xFF})"


# ---- bitwriter.rs / bitreader.rs ----------------------------------------
class BitWriter:
    def __init__(self):
        self.h = lib().zo_bw_new()

    def write_bit(self, bit):
        lib().zo_bw_write_bit(self.h, bit)

    def write_bits(self, value, n):
        lib().zo_bw_write_bits(self.h, value, n)

    def write_byte(self, b):
        lib().zo_bw_write_byte(self.h, b)

    def write_bytes(self, bs):
        lib().zo_bw_write_bytes(self.h, bytes(bs), len(bs))

    def flush_byte(self):
        lib().zo_bw_flush_byte(self.h)

    def finish(self) -> bytes:
        p, n = _u8p(), C.c_size_t()
        lib().zo_bw_finish(self.h, C.byref(p), C.byref(n))
        return _take_bytes(p, n)

    def __del__(self):
        if getattr(self, "h", None):
            lib().zo_bw_free(self.h)
            self.h = None


class BitReader:
    def __init__(self, mem: bytes):
        self.mem = bytes(mem)  # keep alive: the reader borrows it
        self.h = lib().zo_br_new(self.mem, len(self.mem))

    def read_byte(self):
        return lib().zo_br_read_byte(self.h)

    def read_bit(self):
        return lib().zo_br_read_bit(self.h)

    def read_bits(self, n):
        return lib().zo_br_read_bits(self.h, n)

    def read_bytes(self, n):
        return lib().zo_br_read_bytes(self.h, n)

    def __del__(self):
        if getattr(self, "h", None):
            lib().zo_br_free(self.h)
            self.h = None


def code_to_bytes(code, length) -> bytes:
    p, n = _u8p(), C.c_size_t()
    lib().zo_code_to_bytes(code, length, C.byref(p), C.byref(n))
    return _take_bytes(p, n)


def adler32(data: bytes) -> int:
    return lib().zo_adler32(bytes(data), len(data))


def table(name):
    n = {"code_length_codes_order": 19, "length_extra_bits": 29, "length_base": 29,
         "distance_extra_bits": 30, "distance_base": 30}[name]
    p = getattr(lib(), "zo_" + name)()
    return [p[i] for i in range(n)]


def get_length_code(length):
    return lib().zo_get_length_code(length)


def get_distance_code(distance):
    return lib().zo_get_distance_code(distance)


# ---- huffman.rs ----------------------------------------------------------
class Tree:
    def __init__(self, h=None):
        self.h = h if h is not None else lib().zo_tree_new()

    @staticmethod
    def _sym(s):
        return ord(s) if isinstance(s, str) else int(s)

    def insert(self, code, length, symbol):
        lib().zo_tree_insert(self.h, code, length, self._sym(symbol))

    def assign(self):
        lib().zo_tree_assign(self.h)

    def encode(self, symbol):
        c, l = C.c_size_t(), C.c_size_t()
        if lib().zo_tree_encode(self.h, self._sym(symbol), C.byref(c), C.byref(l)):
            return (c.value, l.value)
        return None

    def n_codes(self):
        return lib().zo_tree_n_codes(self.h)

    def max_code_len(self):
        return lib().zo_tree_max_code_len(self.h)

    def to_canonical(self):
        h = lib().zo_tree_to_canonical(self.h)
        if not h:
            raise OracleError(ERR_ARG, "to_canonical before assign (reference panics)")
        return Tree(h)

    def encodings(self):
        n = lib().zo_tree_encodings(self.h, None, None, None, 0)
        s, c, l = (C.c_uint32 * max(1, n))(), (C.c_size_t * max(1, n))(), (C.c_size_t * max(1, n))()
        lib().zo_tree_encodings(self.h, s, c, l, n)
        return {int(s[i]): (int(c[i]), int(l[i])) for i in range(n)}

    def decode(self, reader: BitReader):
        v = lib().zo_tree_decode(self.h, reader.h)
        if v == -2:
            raise OracleError(ERR_MALFORMED, "reader out of bounds (reference panics)")
        return None if v < 0 else v

    @staticmethod
    def from_bitlen_list(bitlen, alphabet):
        a = (C.c_uint32 * max(1, len(alphabet)))(*[Tree._sym(x) for x in alphabet])
        return Tree(lib().zo_tree_from_bitlen_list(_sz_array(list(bitlen)), len(bitlen), a, len(alphabet)))

    @staticmethod
    def from_data(data: bytes, tie=TIE_FIFO, seed=0):
        return Tree(lib().zo_tree_from_data(bytes(data), len(data), tie, seed))

    @staticmethod
    def from_freq(freq: dict, tie=TIE_FIFO, seed=0):
        items = [(Tree._sym(k), v) for k, v in freq.items()]
        s = (C.c_uint32 * max(1, len(items)))(*[k for k, _ in items])
        return Tree(lib().zo_tree_from_freq(s, _sz_array([v for _, v in items]), len(items), tie, seed))

    @staticmethod
    def get_zlib_fixed():
        a, b = C.c_void_p(), C.c_void_p()
        lib().zo_tree_get_zlib_fixed(C.byref(a), C.byref(b))
        return Tree(a), Tree(b)

    @staticmethod
    def from_lz77(tokens, tie=TIE_FIFO, seed=0):
        a, b = C.c_void_p(), C.c_void_p()
        rc = lib().zo_tree_from_lz77(_tok_array(tokens), len(tokens), tie, seed, C.byref(a), C.byref(b))
        if rc:
            raise OracleError(rc)
        return Tree(a), Tree(b)

    @staticmethod
    def decode_trees(reader: BitReader):
        a, b = C.c_void_p(), C.c_void_p()
        rc = lib().zo_tree_decode_trees(reader.h, C.byref(a), C.byref(b))
        if rc:
            raise OracleError(rc)
        return Tree(a), Tree(b)

    def __del__(self):
        if getattr(self, "h", None):
            lib().zo_tree_free(self.h)
            self.h = None


# ---- lz77.rs ------------------------------------------------------------
def lz77_compress(data: bytes, start=0, window=144, min_len=5, max_len=100, literal_compare=False):
    """LZ77Compressor::compress with the crate's default parameters."""
    p, n = _u32p(), C.c_size_t()
    fn = lib().zo_lz77_compress_literal if literal_compare else lib().zo_lz77_compress
    rc = fn(bytes(data), len(data), start, window, min_len, max_len, C.byref(p), C.byref(n))
    if rc:
        raise OracleError(rc)
    return _take_list(p, n.value)


def lz77_zlib(data: bytes, start=0, fast=False):
    """Tokens with get_zlib_compressor's parameters (32768 / 3 / 258)."""
    if fast:
        p, n = _u32p(), C.c_size_t()
        rc = lib().zo_lz77_fast(bytes(data), len(data), start, 32768, 258, C.byref(p), C.byref(n))
        if rc:
            raise OracleError(rc)
        return _take_list(p, n.value)
    return lz77_compress(data, start, 32768, 3, 258)


def lz77_fast(data: bytes, start=0, window=32768, max_len=258):
    p, n = _u32p(), C.c_size_t()
    rc = lib().zo_lz77_fast(bytes(data), len(data), start, window, max_len, C.byref(p), C.byref(n))
    if rc:
        raise OracleError(rc)
    return _take_list(p, n.value)


def detokenize(tokens, prefix: bytes = b"") -> bytes:
    """Expands a token stream (python helper, not part of the reference)."""
    out = bytearray(prefix)
    for t in tokens:
        if t & TOK_MARKER:
            ln, d = (t >> 16) & 0x1FF, t & 0xFFFF
            for _ in range(ln):
                out.append(out[-d])
        else:
            out.append(t & 0xFF)
    return bytes(out[len(prefix):])


# ---- compress.rs pieces ---------------------------------------------------
def _check(rc):
    if rc:
        raise OracleError(rc)


def literal_writer(ltree, w, sym):
    _check(lib().zo_literal_writer(ltree.h, w.h, Tree._sym(sym)))


def length_writer(ltree, w, length):
    _check(lib().zo_length_writer(ltree.h, w.h, length))


def distance_writer(dtree, w, distance):
    _check(lib().zo_distance_writer(dtree.h, w.h, distance))


def write_compressed_data(w, tokens, ltree, dtree):
    _check(lib().zo_write_compressed_data(w.h, _tok_array(tokens), len(tokens), ltree.h, dtree.h))


def write_dynamic_header(w, ltree, dtree, strict=True):
    _check(lib().zo_write_dynamic_header(w.h, ltree.h, dtree.h, int(strict)))


def get_codelengths(ltree, dtree):
    a, b, na, nb = _szp(), _szp(), C.c_size_t(), C.c_size_t()
    _check(lib().zo_get_codelengths(ltree.h, dtree.h, C.byref(a), C.byref(na), C.byref(b), C.byref(nb)))
    return _take_list(a, na.value), _take_list(b, nb.value)


def _pairs(s, e, n):
    syms, ext = _take_list(s, n.value), _take_list(e, n.value)
    return [(a, None if b < 0 else b) for a, b in zip(syms, ext)]


def zlib_rle(cl1, cl2):
    s, e, n = _szp(), _lp(), C.c_size_t()
    _check(lib().zo_zlib_rle(_sz_array(cl1), len(cl1), _sz_array(cl2), len(cl2), C.byref(s), C.byref(e), C.byref(n)))
    return _pairs(s, e, n)


def run_length_encode(data):
    """Returns [('Once', v)] / [('Repeat', v, count)] like the reference enum."""
    v, c, m = _szp(), _szp(), C.c_size_t()
    _check(lib().zo_run_length_encode(_sz_array(data), len(data), C.byref(v), C.byref(c), C.byref(m)))
    vals, cnts = _take_list(v, m.value), _take_list(c, m.value)
    return [("Once", a) if b == 0 else ("Repeat", a, b) for a, b in zip(vals, cnts)]


def rle_to_zlib_codes(rle):
    vals = [r[1] for r in rle]
    cnts = [0 if r[0] == "Once" else r[2] for r in rle]
    s, e, n = _szp(), _lp(), C.c_size_t()
    _check(lib().zo_rle_to_zlib_codes(_sz_array(vals), _sz_array(cnts), len(rle), C.byref(s), C.byref(e), C.byref(n)))
    return _pairs(s, e, n)


def zlib_rle_encode_zero(repetitions):
    s, e, n = _szp(), _lp(), C.c_size_t()
    _check(lib().zo_zlib_rle_encode_zero(repetitions, C.byref(s), C.byref(e), C.byref(n)))
    return _pairs(s, e, n)


def zlib_rle_encode_nonzero(num, repetitions):
    s, e, n = _szp(), _lp(), C.c_size_t()
    _check(lib().zo_zlib_rle_encode_nonzero(num, repetitions, C.byref(s), C.byref(e), C.byref(n)))
    return _pairs(s, e, n)


def get_code_tree(combined, strict=True):
    syms = [c for c, _ in combined]
    t = C.c_void_p()
    _check(lib().zo_get_code_tree(_sz_array(syms), len(syms), int(strict), C.byref(t)))
    return Tree(t)


def get_hcodes(code_tree):
    p, n = _szp(), C.c_size_t()
    rc = lib().zo_get_hcodes(code_tree.h, C.byref(p), C.byref(n))
    out = _take_list(p, n.value)
    _check(rc)
    return out


# ---- whole streams ---------------------------------------------------------
def compress_reference(data: bytes, strategy: int) -> bytes:
    """compress(&data, &strategy) exactly as the crate would (or OracleError
    carrying the PANIC_* code where the crate panics)."""
    p, n = _u8p(), C.c_size_t()
    _check(lib().zo_compress_reference(bytes(data), len(data), strategy, C.byref(p), C.byref(n)))
    return _take_bytes(p, n)


def compress_chunked(data: bytes, strategy: int, chunk_bytes: int = 1 << 20, fast: bool = True) -> bytes:
    p, n = _u8p(), C.c_size_t()
    _check(lib().zo_compress_chunked(bytes(data), len(data), strategy, chunk_bytes, int(fast), C.byref(p), C.byref(n)))
    return _take_bytes(p, n)


def chunk_tokens(data: bytes, chunk_bytes: int, index: int, fast: bool = True):
    p, n = _u32p(), C.c_size_t()
    _check(lib().zo_chunk_tokens(bytes(data), len(data), chunk_bytes, index, int(fast), C.byref(p), C.byref(n)))
    return _take_list(p, n.value)


def chunk_tables(tokens):
    ll_len, d_len = (C.c_uint8 * 286)(), (C.c_uint8 * 30)()
    ll_code, d_code = (C.c_uint16 * 286)(), (C.c_uint16 * 30)()
    _check(lib().zo_chunk_tables(_tok_array(tokens), len(tokens), ll_len, ll_code, d_len, d_code))
    return list(ll_len), list(ll_code), list(d_len), list(d_code)


def block_bits(tokens, strategy=DYNAMIC) -> int:
    b = C.c_uint64()
    _check(lib().zo_block_bits(_tok_array(tokens), len(tokens), strategy, C.byref(b)))
    return b.value


def write_block(tokens, strategy=DYNAMIC, bfinal=True):
    """(bytes, bit length) of the single block the build emits for `tokens`."""
    p, n, b = _u8p(), C.c_size_t(), C.c_uint64()
    _check(lib().zo_write_block(_tok_array(tokens), len(tokens), strategy, int(bfinal), C.byref(p), C.byref(n), C.byref(b)))
    return _take_bytes(p, n), b.value


def auto_choice(data: bytes) -> int:
    return lib().zo_auto_choice(bytes(data), len(data))


def decompress(data: bytes) -> bytes:
    p, n = _u8p(), C.c_size_t()
    err = C.create_string_buffer(256)
    rc = lib().zo_decompress(bytes(data), len(data), C.byref(p), C.byref(n), err, 256)
    if rc:
        raise OracleError(rc, err.value.decode())
    return _take_bytes(p, n)


def bench_slices(data: bytes, slices, threads: int):
    """slices: list of (offset, length, halo). Returns compressed bits per slice."""
    n = len(slices)
    bits = (C.c_uint64 * max(1, n))()
    rc = lib().zo_bench_slices(data, _sz_array([s[0] for s in slices]), _sz_array([s[1] for s in slices]),
                               _sz_array([s[2] for s in slices]), n, threads, bits)
    _check(rc)
    return [int(bits[i]) for i in range(n)]
