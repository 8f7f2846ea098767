#!/usr/bin/env python3
"""Times zg_decompress_blocks_device on a corpus compressed by the library: python tools/time_inflate.py KIND MIB [REPS]."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import zlib_amd as za  # noqa: E402

kind, mib = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
n, chunk = mib << 20, 1 << 20
ctx = za.Context(0)
d_in = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
host = torch.empty(min(n, 64 << 20), dtype=torch.uint8).pin_memory()
for at in range(0, n, host.numel()):
    m = min(host.numel(), n - at)
    za.corpus_into(kind, host.data_ptr(), m, seed=0x5EED0004, offset=at)
    d_in[at:at + m].copy_(host[:m])
cap = za.compress_bound(n, za.Strategy.Dynamic, chunk_bytes=chunk)
d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
out_n = za.compress_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, za.Strategy.Dynamic, chunk_bytes=chunk, ctx=ctx)
offs = za.last_block_offsets(ctx=ctx)
back = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
for r in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    za.decompress_blocks_device(d_out.data_ptr(), out_n, offs, chunk, back.data_ptr(), n, ctx=ctx)
    torch.cuda.synchronize()
    print(f"{kind} {mib} MiB: inflate {1e3 * (time.perf_counter() - t0):.1f} ms, ratio {out_n / n:.3f}", flush=True)
assert torch.equal(back[:n], d_in[:n])
