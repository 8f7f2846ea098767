"""The multi-rank assembly logic (zlib_amd/sharding.py) over gloo on the CPU,
world size 2 and 3.  The per-rank compute is replaced by a stand-in engine that
produces, from the oracle, exactly what zg_shard_begin / zg_shard_finish return
on a GPU; what is tested is the exchange: bit totals and Adler parts gathered,
bit phases, the shared boundary byte, the payload sends."""
import os
import socket
import zlib as sysz

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import zref
from zlib_amd import _lib, sharding


def _corpus(n):
    import zlib_amd
    return zlib_amd.corpus("text", n, seed=11)


class OracleEngine:
    def __init__(self, data, start, n, chunk):
        self.data, self.start, self.n, self.chunk = data, start, n, chunk
        self.out = torch.zeros(n * 2 + 4096, dtype=torch.uint8)
        self.blocks = []

    def begin(self, last_rank):
        nchunks = max(1, -(-self.n // self.chunk))
        c0 = self.start // self.chunk
        self.blocks = []
        for c in range(nchunks):
            toks = zref.chunk_tokens(self.data, self.chunk, c0 + c, fast=True)
            self.blocks.append(zref.write_block(toks, zref.DYNAMIC, last_rank and c == nchunks - 1))
        return sum(b for _, b in self.blocks), zref.adler32(self.data[self.start:self.start + self.n]), self.n

    def finish(self, base_bit, first_rank, last_rank, adler):
        pos = 16 if first_rank else base_bit % 8
        acc = 0x0178 if first_rank else 0
        for blk, bits in self.blocks:
            acc |= int.from_bytes(blk, "little") << pos
            pos += bits
        nbytes = (pos + 7) // 8
        raw = acc.to_bytes(nbytes, "little")
        if last_rank:
            raw += adler.to_bytes(4, "big")
        self.out[:len(raw)] = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
        return len(raw)


def _worker(rank, world, port, total, chunk, q, piece=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if piece:
        sharding.P2P_PIECE = piece  # payloads travel in several point-to-point operations
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = _corpus(total)
        start, n, _halo = sharding.plan_shards(total, chunk, world)[rank]
        eng = OracleEngine(data, start, n, chunk)
        final = torch.zeros(total * 2 + 4096, dtype=torch.uint8) if rank == 0 else None
        for _ in range(2):  # twice: scratch tensors are reused
            length = sharding.compress_sharded(eng, dist, torch, rank, world, torch.device("cpu"), final=final)
        if rank == 0:
            q.put(bytes(final[:length].numpy().tobytes()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,total,chunk,piece", [(2, 150_000, 16384, None), (2, 65536, 32768, None),
                                                     (3, 100_000, 8192, None), (3, 100_000, 8192, 1000)])
def test_sharded_assembly_equals_single_stream(world, total, chunk, piece):
    _lib.lib()  # adler32_combine comes from the product library (host function)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, chunk, q, piece)) for r in range(world)]
    for p in procs:
        p.start()
    stream = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    data = _corpus(total)
    assert stream == zref.compress_chunked(data, zref.DYNAMIC, chunk, fast=True)
    assert sysz.decompress(stream) == data


def test_plan_shards_covers_input():
    for total, chunk, world in [(0, 1024, 2), (1000, 1024, 4), (10_000_000, 1 << 20, 8), (3 << 20, 1 << 20, 2)]:
        plan = sharding.plan_shards(total, chunk, world)
        assert plan[0][0] == 0 and sum(n for _, n, _ in plan) == total
        for (s0, n0, _), (s1, _, h1) in zip(plan, plan[1:]):
            assert s0 + n0 == s1 and s1 % chunk == 0 or s1 == total
            assert h1 == min(32768, s1)


def test_adler32_combine_host():
    import random
    L = _lib.lib()
    rng = random.Random(5)
    for _ in range(50):
        a = bytes(rng.randrange(256) for _ in range(rng.randrange(0, 3000)))
        b = bytes(rng.randrange(256) for _ in range(rng.randrange(0, 3000)))
        assert L.zg_adler32_combine(sysz.adler32(a), sysz.adler32(b), len(b)) == sysz.adler32(a + b)
    big = 7 << 30
    assert L.zg_adler32_combine(1, 1, big) == 1
