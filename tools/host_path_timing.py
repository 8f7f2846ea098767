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
# the stream front end: the same bytes pulled from / pushed to files (segments of 256 MiB, three threads)
import tempfile
with tempfile.TemporaryDirectory() as td:
    src, dst = os.path.join(td, "in.bin"), os.path.join(td, "out.z")
    with open(src, "wb") as f:
        f.write(data)
    for rep in range(2):
        t0 = time.perf_counter()
        n_in, n_out = za.compress_file(src, dst, za.Strategy.Dynamic)
        dt = time.perf_counter() - t0
        print(f"file to file: {n_in / dt / 1e9:.2f} GB/s ({dt * 1e3:.0f} ms)", flush=True)
    with open(dst, "rb") as f:
        print("file stream equals the in-memory one", f.read() == out)
