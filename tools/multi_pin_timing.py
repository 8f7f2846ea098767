#!/usr/bin/env python3
"""zg_compress_multi on host buffers with and without page-locking them for the call (ZLIB_AMD_MULTI_PIN), the one GPU
of the box standing in for two devices: python tools/multi_pin_timing.py [MiB]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zlib_amd as za  # noqa: E402

n = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20
data = za.corpus("mixed", n, seed=0x5EED0004)
for pin in ("0", "1", "0", "1"):
    os.environ["ZLIB_AMD_MULTI_PIN"] = pin
    t0 = time.perf_counter()
    out = za.compress_multi(data, [0, 0], za.Strategy.Dynamic)
    dt = time.perf_counter() - t0
    print(f"ZLIB_AMD_MULTI_PIN={pin}: {n >> 20} MiB over two runs in {dt * 1e3:.0f} ms ({n / dt / 1e9:.2f} GB/s), {len(out)} bytes", flush=True)
