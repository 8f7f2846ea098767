#!/usr/bin/env python3
"""configs[0] of the baseline: compress + decompress 1 MiB of synthetic text, Strategy::Fixed, on the CPU with the
oracle's restatement of the reference (verbatim exhaustive matcher) -- round-trip plumbing, no GPU involved."""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import zref as z
import zlib_amd as za  # host-side corpus generator only

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
data = za.corpus("text", n, seed=0x5EED0001)
t0 = time.perf_counter()
stream = z.compress_reference(data, z.FIXED)
t1 = time.perf_counter()
back = z.decompress(stream)
t2 = time.perf_counter()
print(f"{n} bytes -> {len(stream)} ({len(stream) / n:.4f}); compress {t1 - t0:.1f} s ({n / (t1 - t0) / 1e6:.3f} MB/s), "
      f"decompress {t2 - t1:.3f} s; oracle round trip {back == data}, system zlib {zlib.decompress(stream) == data}")
