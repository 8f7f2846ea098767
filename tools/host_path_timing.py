#!/usr/bin/env python3
"""Times zg_compress with host buffers in and out (the path a crate user takes) beside the device-resident one."""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zlib_amd as za
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
data = za.corpus("mixed", n, seed=0x5EED0004)
za.compress(data[:1 << 20], za.Strategy.Dynamic)  # context, code objects
for rep in range(3):
    t0 = time.perf_counter()
    out = za.compress(data, za.Strategy.Dynamic)
    dt = time.perf_counter() - t0
    print(f"host in / host out: {n / dt / 1e9:.2f} GB/s ({dt * 1e3:.0f} ms), ratio {len(out) / n:.4f}", flush=True)
ok = zlib.decompress(out) == data
print("round trip", ok)
