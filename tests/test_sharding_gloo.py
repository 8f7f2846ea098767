"""The multi-rank assembly logic (zlib_amd/sharding.py) over gloo on the CPU,
world size 2, 3, 4 and 8.  The per-rank compute is replaced by a stand-in engine that
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
        nchunks = -(-self.n // self.chunk) or (1 if last_rank else 0)  # a run without bytes adds nothing unless it is the last
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


class OraclePieceEngine:
    """What zg_shard_open / _next / _merge / _seal do on a GPU, restated with the oracle's blocks and Python integers:
    a pass is `pass_chunks` blocks; after it the buffer holds all bits so far, of which the complete bytes are final."""

    def __init__(self, data, start, n, chunk, pass_chunks, fail_first_run=False):
        self.data, self.start, self.n, self.chunk, self.pass_chunks = data, start, n, chunk, pass_chunks
        self.packed = torch.zeros(n * 2 + 4096, dtype=torch.uint8)
        self.fail = fail_first_run  # the first run ends as a run whose device-side self-check failed

    def open(self, last_rank):
        nchunks = -(-self.n // self.chunk) or (1 if last_rank else 0)  # a run without bytes adds nothing unless it is the last
        c0 = self.start // self.chunk
        self.blocks = []
        for c in range(nchunks):
            toks = zref.chunk_tokens(self.data, self.chunk, c0 + c, fast=True)
            self.blocks.append(zref.write_block(toks, zref.DYNAMIC, last_rank and c == nchunks - 1))
        self.acc = self.pos = self.at = self.reported = 0
        self.packed.fill_(0xEE)  # stale bytes of an earlier call

    def next(self):
        assert self.at < max(1, len(self.blocks))
        for blk, bits in self.blocks[self.at:self.at + self.pass_chunks]:
            self.acc |= int.from_bytes(blk, "little") << self.pos
            self.pos += bits
        self.at = min(len(self.blocks), self.at + self.pass_chunks)
        last = self.at == len(self.blocks)
        if last and self.fail:
            self.fail = False
            raise _lib.ZgError(_lib.ERR_INTERNAL, "self-check")
        raw = self.acc.to_bytes((self.pos + 7) // 8, "little")  # the unfinished byte is in the buffer, not in the piece
        if len(raw) > self.reported:
            self.packed[self.reported:len(raw)] = torch.frombuffer(bytearray(raw[self.reported:]), dtype=torch.uint8)
        done = len(raw) if last else self.pos // 8
        off, self.reported = self.reported, done
        ad = zref.adler32(self.data[self.start:self.start + self.n]) if last else None
        return off, done - off, self.pos, last, ad

    @staticmethod
    def merge(src, n_bits, final, base_bit, first_run):
        run = int.from_bytes(src[:(n_bits + 7) // 8].numpy().tobytes(), "little") & ((1 << n_bits) - 1)
        if first_run:
            assert base_bit == 16
            final[0], final[1] = 0x78, 0x01
        lo, ph = base_bit // 8, base_bit % 8
        keep = int(final[lo]) & ((1 << ph) - 1) if ph else 0
        raw = (keep | (run << ph)).to_bytes((ph + n_bits + 7) // 8, "little")
        if raw:
            final[lo:lo + len(raw)] = torch.frombuffer(bytearray(raw), dtype=torch.uint8)

    @staticmethod
    def seal(final, end_bit, adler):
        at = (end_bit + 7) // 8
        final[at:at + 4] = torch.frombuffer(bytearray(adler.to_bytes(4, "big")), dtype=torch.uint8)
        return at + 4


def _piece_worker(rank, world, port, total, chunk, pass_chunks, q, piece=None, fail_rank=-1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if piece:
        sharding.P2P_PIECE = piece
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = _corpus(total)
        plan = sharding.plan_shards(total, chunk, world)
        start, n, _halo = plan[rank]
        eng = OraclePieceEngine(data, start, n, chunk, pass_chunks, fail_first_run=rank == fail_rank)
        final = stages = None
        if rank == 0:
            final = torch.full((total * 2 + 4096,), 0xEE, dtype=torch.uint8)
            stages = [None] + [torch.full((plan[r][1] * 2 + 4096,), 0xEE, dtype=torch.uint8) for r in range(1, world)]
        scratch = {}
        for _ in range(2):  # twice: scratch tensors and staging buffers are reused
            length = sharding.compress_sharded_overlapped(eng, dist, torch, rank, world, torch.device("cpu"),
                                                          final=final, stages=stages, scratch=scratch)
        if rank == 0:
            q.put(bytes(final[:length].numpy().tobytes()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,total,chunk,pass_chunks,piece", [
    (2, 150_000, 16384, 2, None),   # 5 chunks each way: ranks with 2 and 3 chunks, 1 and 2 passes
    (2, 65536, 32768, 4, None),     # one pass per rank
    (3, 100_000, 8192, 1, None),    # a pass per chunk: 4 or 5 passes per rank
    (3, 100_000, 8192, 3, 1000),    # pieces travel in several point-to-point operations
    (3, 20_000, 8192, 1, None),     # 3 chunks over 3 ranks, the last one short
    (3, 10_000, 8192, 2, None),     # 2 chunks over 3 ranks: the first rank has no input and adds nothing
    (2, 0, 8192, 2, None),          # no input at all: the last rank writes the one empty block
    (4, 130_000, 8192, 2, None),    # 16 chunks over 4 ranks, two passes each
    (8, 200_000, 8192, 1, 3000),    # as many ranks as a node has GPUs: 25 chunks, 3 or 4 a rank, pieces in several sends
])
def test_overlapped_assembly_equals_single_stream(world, total, chunk, pass_chunks, piece):
    """The exchange that runs beside the kernels: per pass one all_gather of piece sizes and one send of the bytes
    that pass made final; rank 0 merges the staged runs at their bit offsets when all totals are in."""
    _lib.lib()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_piece_worker, args=(r, world, port, total, chunk, pass_chunks, q, piece))
             for r in range(world)]
    for p in procs:
        p.start()
    stream = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    data = _corpus(total)
    assert stream == zref.compress_chunked(data, zref.DYNAMIC, chunk, fast=True)
    assert sysz.decompress(stream) == data


@pytest.mark.parametrize("fail_rank", [0, 1, 2])
def test_overlapped_assembly_starts_over_when_a_rank_fails_its_self_check(fail_rank):
    """ZG_ERR_INTERNAL from a rank's last zg_shard_next (the library has switched to its conservative kernels): the
    rank reports it with its last row, every rank drops what it has and runs again; rank 1 has fewer passes than the
    others, so ranks that are done and ranks in the middle of their run are both among those that start over."""
    _lib.lib()
    world, total, chunk, pass_chunks = 3, 100_000, 8192, 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_piece_worker, args=(r, world, port, total, chunk, pass_chunks, q, None, fail_rank))
             for r in range(world)]
    for p in procs:
        p.start()
    stream = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert stream == zref.compress_chunked(_corpus(total), zref.DYNAMIC, chunk, fast=True)


def test_overlapped_assembly_with_one_rank():
    """sharding.SingleRank: the same calls with nothing to exchange."""
    _lib.lib()
    total, chunk = 70_000, 8192
    data = _corpus(total)
    eng = OraclePieceEngine(data, 0, total, chunk, 2)
    final = torch.full((total * 2 + 4096,), 0xEE, dtype=torch.uint8)
    n = sharding.compress_sharded_overlapped(eng, sharding.SingleRank, torch, 0, 1, torch.device("cpu"), final=final,
                                             stages=[None])
    assert bytes(final[:n].numpy().tobytes()) == zref.compress_chunked(data, zref.DYNAMIC, chunk, fast=True)


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
                                                     (3, 100_000, 8192, None), (3, 100_000, 8192, 1000),
                                                     (3, 10_000, 8192, None)])  # the last: a rank without input
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
