#!/usr/bin/env python3
"""Steps and entries of the two level-2 passes by position modulo 16 (library built with -DZG_DEBUG_COUNTERS)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import zlib_amd as za
from zlib_amd import _lib
kind = sys.argv[1] if len(sys.argv) > 1 else "binary"
n = 64 << 20
data = za.corpus(kind, n, seed=0x5EED0004)
out = za.compress(data, za.Strategy.Dynamic)
L = _lib.lib()
buf = (C.c_ulonglong * 64)()
L.zg_debug_counters(buf)
v = list(buf)
for name, off in (("pass1 steps", 0), ("pass2 steps", 16), ("pass1 entries", 32), ("pass2 entries", 48)):
    print(name, [x for x in v[off:off + 16]], "total", sum(v[off:off + 16]))
print("positions", n, "ratio", len(out) / n)
