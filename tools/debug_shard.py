#!/usr/bin/env python3
"""Rehearsal of the sharded path over gloo with per-rank checksums of the payload (run under torch.distributed.run):
tells a transport fault from a fault in a rank's own bytes."""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import zlib_amd as za
from zlib_amd import sharding
import bench

total = int(sys.argv[1]) if len(sys.argv) > 1 else 8 << 30
kind, seed, chunk = "mixed", 0x5EED0004, 1 << 20
world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
dev_index = rank % torch.cuda.device_count()
torch.cuda.set_device(dev_index)
device = torch.device("cuda", dev_index)
dist.init_process_group("gloo")
ctx = za.Context(dev_index)
start, n, halo = sharding.plan_shards(total, chunk, world)[rank]
d_in = bench.upload_corpus(torch, za, kind, seed, start - halo, halo + n, device)
cap = za.compress_bound(n, za.Strategy.Dynamic, chunk_bytes=chunk) + 8
d_out = torch.empty(cap, dtype=torch.uint8, device=device)
final = torch.empty(za.compress_bound(total, za.Strategy.Dynamic, chunk_bytes=chunk) + 8, dtype=torch.uint8, device=device) if rank == 0 else None
engine = sharding.HipEngine(ctx, d_in.data_ptr(), halo, n, za.Strategy.Dynamic, chunk, d_out)
# the steps of compress_sharded, with the layout kept
bits, ad, nbytes = engine.begin(rank == world - 1)
infos = [None] * world
dist.all_gather_object(infos, (bits, ad, nbytes))
bases, end_bit, adler = sharding.stream_layout(infos)
nb = engine.finish(bases[rank], rank == 0, rank == world - 1, adler)
mine = d_out[:nb].cpu().numpy()
crc_body = zlib.crc32(memoryview(mine)[1:])
scratch = {}
length = sharding.compress_sharded(engine, dist, torch, rank, world, device, final=final, scratch=scratch)
crcs = [None] * world
dist.all_gather_object(crcs, (nb, int(mine[0]), crc_body))
if rank == 0:
    total_bytes = (end_bit + 7) // 8 + 4
    print("edges received", scratch["edges"].cpu().tolist(), "first bytes sent", [c[1] for c in crcs], flush=True)
    print("bases", bases, "end_bit", end_bit, "total_bytes", total_bytes, "length", length, flush=True)
    host = final[:length].cpu().numpy()
    for r in range(world):
        off = bases[r] // 8
        stop = total_bytes if r == world - 1 else (bases[r + 1] + 7) // 8
        got = zlib.crc32(memoryview(host)[off + 1:stop])
        print("rank", r, "nb", crcs[r][0], "expected span", stop - off, "crc sent", crcs[r][2], "crc in final", got,
              "MATCH" if got == crcs[r][2] and crcs[r][0] == stop - off else "DIFFER", flush=True)
    # the same corpus through the single-GPU path: the two streams must be equal byte for byte
    whole = bench.upload_corpus(torch, za, kind, seed, 0, total, device)
    capw = za.compress_bound(total, za.Strategy.Dynamic, chunk_bytes=chunk)
    single = torch.empty(capw, dtype=torch.uint8, device=device)
    ln = za.compress_device(whole.data_ptr(), total, single.data_ptr(), capw, za.Strategy.Dynamic, chunk_bytes=chunk, ctx=ctx)
    print("single length", ln, flush=True)
    m = min(ln, length)
    step = 256 << 20
    first_bad = None
    nbad = 0
    for at in range(0, m, step):
        e = min(m, at + step)
        ne = (final[at:e] != single[at:e])
        c = int(ne.sum().item())
        if c and first_bad is None:
            first_bad = at + int(ne.nonzero()[0].item())
        nbad += c
    print("differing bytes", nbad, "first at", first_bad, flush=True)
    if first_bad is not None:
        lo = max(0, first_bad - 8)
        print("final ", final[lo:first_bad + 24].cpu().tolist(), flush=True)
        print("single", single[lo:first_bad + 24].cpu().tolist(), flush=True)
        print("seam byte", bases[1] // 8, flush=True)
dist.barrier()
dist.destroy_process_group()
