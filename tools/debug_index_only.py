#!/usr/bin/env python3
"""Time of the index kernel alone (library built with -DZG_INDEX_DEBUG in zg_match.hip and zg_api.hip): 256 MiB of each
kind of corpus.  usage: tools/debug_index_only.py [kinds...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import zlib_amd as za
from zlib_amd import _lib
L = _lib.lib()
L.zg_debug_index_only.restype = C.c_double
L.zg_debug_index_only.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
ctx = za.default_context()
n = 256 << 20
out = []
for kind in (sys.argv[1:] or ["mixed", "text", "binary", "random"]):
    d = torch.frombuffer(bytearray(za.corpus(kind, n, seed=0x5EED0004)), dtype=torch.uint8).cuda()
    out.append(f"{kind} {L.zg_debug_index_only(ctx.handle, d.data_ptr(), n, 5):.3f} ms")
print(os.environ.get("TAG", ""), " | ".join(out))
