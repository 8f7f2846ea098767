#!/usr/bin/env python3
"""Throughput of zg_compress_device on 2 GiB of the mixed corpus for several batch sizes (zg_opts.batch_bytes):
flat from 256 MiB upward, which is why that is the default."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import zlib_amd as za, bench
torch.zeros(1, device="cuda")
ctx = za.Context(0)
n = 2 << 30
d_in = bench.upload_corpus(torch, za, "mixed", 0x5EED0004, 0, n, torch.device("cuda", 0))
cap = za.compress_bound(n, za.Strategy.Dynamic)
d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
for bb in (128 << 20, 256 << 20, 512 << 20, 1 << 30, 2 << 30):
    za.compress_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, za.Strategy.Dynamic, batch_bytes=bb, ctx=ctx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        m = za.compress_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, za.Strategy.Dynamic, batch_bytes=bb, ctx=ctx)
    dt = (time.perf_counter() - t0) / 2
    print(f"batch {bb >> 20} MiB: {n / dt / 1e9:.2f} GB/s ({dt * 1e3:.0f} ms), out {m}", flush=True)
