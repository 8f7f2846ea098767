"""ctypes binding of the C ABI declared in include/zlib_amd.h.

The shared library is the product: hand-written HIP kernels for gfx950 behind
plain C entry points.  There is no Python or CPU implementation of the
compress path here; if the library is missing the import fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libzlib_amd.so")

OK = 0
ERR_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_NOMEM, ERR_CAPACITY, ERR_UNSUPPORTED, ERR_INTERNAL, ERR_DATA, ERR_IO = -1, -2, -3, -4, -5, -6, -7, -8, -9

#: every symbol include/zlib_amd.h declares
SYMBOLS = [
    "zg_init", "zg_destroy", "zg_strerror", "zg_last_error", "zg_free", "zg_compress", "zg_compress_bound",
    "zg_compress_device", "zg_compress_multi", "zg_compress_stream", "zg_compress_file", "zg_last_block_offsets", "zg_compress_batch", "zg_compress_batch_bound", "zg_compress_batch_device",
    "zg_decompress", "zg_decompress_batch", "zg_decompress_batch_device", "zg_decompress_blocks_device", "zg_shard_begin", "zg_shard_finish", "zg_shard_open", "zg_shard_next", "zg_shard_merge", "zg_shard_seal", "zg_adler32_combine", "zg_adler32",
    "zg_adler32_device", "zg_adler32_device_noalloc", "zg_lz77", "zg_huffman_from_tokens", "zg_profile_enable", "zg_profile_reset",
    "zg_profile_count", "zg_profile_get", "zg_corpus_fill",
]


class Opts(C.Structure):
    _fields_ = [("chunk_bytes", C.c_size_t), ("batch_bytes", C.c_size_t)]


class ShardInfo(C.Structure):
    _fields_ = [("total_bits", C.c_uint64), ("adler", C.c_uint32), ("n_chunks", C.c_uint32)]


class ShardPiece(C.Structure):
    _fields_ = [("byte_offset", C.c_uint64), ("n_bytes", C.c_uint64), ("bits_so_far", C.c_uint64), ("last", C.c_int)]


#: size_t read_fn(void *user, uint8_t *buf, size_t cap) / int write_fn(void *user, const uint8_t *buf, size_t n)
READ_FN = C.CFUNCTYPE(C.c_size_t, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)
WRITE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)


class ZgError(RuntimeError):
    def __init__(self, code: int, detail: str = ""):
        self.code = code
        msg = lib().zg_strerror(code).decode()
        super().__init__(f"zlib_amd error {code}: {msg}" + (f" ({detail})" if detail else ""))


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950); this package has no fallback path")
        L = C.CDLL(LIB_PATH)
        _declare(L)
        _lib = L
    return _lib


def _declare(L):
    vp, sz, u8p, u32p = C.c_void_p, C.c_size_t, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)

    def f(name, res, *args):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = list(args)

    f("zg_init", C.c_int, C.c_int, C.POINTER(vp))
    f("zg_destroy", None, vp)
    f("zg_strerror", C.c_char_p, C.c_int)
    f("zg_last_error", C.c_char_p, vp)
    f("zg_free", None, vp)
    f("zg_compress", C.c_int, vp, C.c_char_p, sz, C.c_int, C.POINTER(Opts), C.POINTER(u8p), C.POINTER(sz))
    f("zg_compress_bound", sz, sz, C.c_int, C.POINTER(Opts))
    f("zg_compress_device", C.c_int, vp, vp, sz, C.c_int, C.POINTER(Opts), vp, sz, C.POINTER(sz))
    f("zg_compress_multi", C.c_int, C.POINTER(C.c_int), C.c_int, C.c_char_p, sz, C.c_int, C.POINTER(Opts),
      C.POINTER(u8p), C.POINTER(sz))
    f("zg_compress_stream", C.c_int, vp, READ_FN, vp, WRITE_FN, vp, C.c_int, C.POINTER(Opts), C.c_uint64,
      C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))
    f("zg_compress_file", C.c_int, vp, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(Opts), C.POINTER(C.c_uint64),
      C.POINTER(C.c_uint64))
    f("zg_last_block_offsets", sz, vp, C.POINTER(C.c_uint64), sz)
    f("zg_compress_batch", C.c_int, vp, C.c_char_p, C.POINTER(sz), sz, C.c_int, C.POINTER(C.POINTER(C.c_uint8)),
      C.POINTER(C.POINTER(sz)))
    f("zg_compress_batch_bound", sz, sz, sz, C.c_int)
    f("zg_compress_batch_device", C.c_int, vp, vp, sz, vp, sz, C.c_int, vp, sz, vp, C.POINTER(sz))
    f("zg_decompress", C.c_int, vp, C.c_char_p, sz, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(sz))
    f("zg_decompress_batch", C.c_int, vp, C.c_char_p, C.POINTER(sz), sz, C.POINTER(C.POINTER(C.c_uint8)),
      C.POINTER(C.POINTER(sz)))
    f("zg_decompress_batch_device", C.c_int, vp, vp, vp, sz, vp, vp, vp, vp)
    f("zg_decompress_blocks_device", C.c_int, vp, vp, sz, C.POINTER(C.c_uint64), sz, sz, vp, sz)
    f("zg_shard_begin", C.c_int, vp, vp, sz, sz, C.c_int, C.POINTER(Opts), C.c_int, C.POINTER(ShardInfo))
    f("zg_shard_finish", C.c_int, vp, C.c_uint64, C.c_int, C.c_int, C.c_uint32, vp, sz, C.POINTER(sz))
    f("zg_shard_open", C.c_int, vp, vp, sz, sz, C.c_int, C.POINTER(Opts), C.c_int, vp, sz)
    f("zg_shard_next", C.c_int, vp, C.POINTER(ShardPiece), C.POINTER(ShardInfo))
    f("zg_shard_merge", C.c_int, vp, vp, C.c_uint64, vp, sz, C.c_uint64, C.c_int)
    f("zg_shard_seal", C.c_int, vp, vp, sz, C.c_uint64, C.c_uint32, C.POINTER(sz))
    f("zg_adler32_combine", C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64)
    f("zg_adler32", C.c_int, vp, C.c_char_p, sz, C.POINTER(C.c_uint32))
    f("zg_adler32_device", C.c_int, vp, vp, sz, C.POINTER(C.c_uint32))
    f("zg_adler32_device_noalloc", C.c_int, vp, vp, sz, vp)
    f("zg_lz77", C.c_int, vp, C.c_char_p, sz, sz, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(u32p), C.POINTER(sz))
    f("zg_huffman_from_tokens", C.c_int, vp, u32p, sz, u8p, C.POINTER(C.c_uint16), u8p, C.POINTER(C.c_uint16))
    f("zg_profile_enable", C.c_int, vp, C.c_int)
    f("zg_profile_reset", C.c_int, vp)
    f("zg_profile_count", C.c_int, vp)
    f("zg_profile_get", C.c_int, vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_uint64))
    f("zg_corpus_fill", C.c_int, C.c_int, C.c_uint64, C.c_uint64, vp, sz)
