"""Multi-GPU assembly of one zlib stream from per-rank runs of chunks.

The input is cut into contiguous runs of whole chunks, one run per rank
(a chunk's blocks depend only on its own bytes and the 32 KiB of input before them).  The
only exchange is:

  1. one all_gather of three integers per rank (block bit total, Adler-32 of
     the rank's bytes, byte count), from which every rank derives the absolute
     bit offset of its first block and the stream's Adler-32;
  2. a point-to-point send of each rank's packed bytes to rank 0, which ORs
     the single byte two neighbouring ranks share when a rank starts in the
     middle of a byte.

The compute is behind an `engine` object with two calls, so that the same
assembly logic runs over RCCL with the HIP engine (HipEngine) and over gloo on
the CPU with a stand-in engine in the tests.

compress_sharded_overlapped is the same job with the payload exchange running
beside the kernels: every rank compresses its run pass by pass (one pass = the
chunks that fit the scratch buffers), and the bytes a pass has made final
travel to rank 0 while the GPU works on the next pass.  Per pass there is one
all_gather of five integers per rank (the piece's byte count, the bits so far,
the Adler-32 once known, a flag, the rank's input bytes) -- it tells rank 0
what to receive -- and one point-to-point send per rank.  Rank 0 keeps every
rank's bytes in a staging buffer of that rank's (they start at bit 0 there: where
a run lands in the stream is only known when all runs before it are complete)
and, once all totals are in, moves each run to its bit offset with one pass of
a device kernel over the compressed bytes.  What is left on the critical path
after the last pass's kernels: that pass's piece on the wire, and the merge.
"""
from __future__ import annotations

from typing import List, Tuple

from . import _lib, api

ADLER_INIT = 1
P2P_PIECE = 1 << 30  # bytes per point-to-point operation (some backends count a message in 32 bits)
ZLIB_HEADER_BITS = 16


def plan_shards(total_bytes: int, chunk_bytes: int, world: int) -> List[Tuple[int, int, int]]:
    """(start, n, halo) per rank: contiguous runs of whole chunks (the last run takes the ragged tail)."""
    n_chunks = max(1, -(-total_bytes // chunk_bytes))
    out = []
    for r in range(world):
        c0, c1 = (r * n_chunks) // world, ((r + 1) * n_chunks) // world
        start = min(total_bytes, c0 * chunk_bytes)
        end = min(total_bytes, c1 * chunk_bytes)
        out.append((start, end - start, min(32768, start)))
    return out


def stream_layout(infos: List[Tuple[int, int, int]]):
    """From the gathered (total_bits, adler, n_bytes) of every rank: first-block bit offset per rank, the
    end bit of the last block, and the Adler-32 of the whole input."""
    base, adler, bases = ZLIB_HEADER_BITS, ADLER_INIT, []
    for bits, ad, nbytes in infos:
        bases.append(base)
        base += bits
        adler = api.adler32_combine(adler, ad, nbytes)
    return bases, base, adler


class HipEngine:
    """The product engine: zg_shard_begin / zg_shard_finish on this rank's GPU."""

    def __init__(self, ctx, d_in_ptr: int, halo: int, n: int, strategy, chunk_bytes: int, out_tensor):
        self.ctx, self.d_in, self.halo, self.n = ctx, d_in_ptr, halo, n
        self.strategy, self.chunk_bytes, self.out = strategy, chunk_bytes, out_tensor

    def begin(self, last_rank: bool):
        info = api.shard_begin(self.d_in, self.halo, self.n, self.strategy, last_rank, chunk_bytes=self.chunk_bytes,
                               ctx=self.ctx)
        return info.total_bits, info.adler, self.n

    def finish(self, base_bit: int, first_rank: bool, last_rank: bool, adler: int) -> int:
        return api.shard_finish(base_bit, first_rank, last_rank, adler, self.out.data_ptr(), self.out.numel(),
                                ctx=self.ctx)


def compress_sharded(engine, dist, torch, rank: int, world: int, device, final=None, scratch=None):
    """One sharded compress.  Returns the stream length in bytes; on rank 0 `final[:length]` holds the
    stream.  `engine.out` is the rank's own output tensor (uint8)."""
    first, last = rank == 0, rank == world - 1
    bits, ad, nbytes = engine.begin(last)
    if scratch is None:
        scratch = {}
    local = scratch.setdefault("local", torch.zeros(3, dtype=torch.int64, device=device))
    gathered = scratch.setdefault("all", torch.zeros(3 * world, dtype=torch.int64, device=device))
    edge = scratch.setdefault("edge", torch.zeros(1, dtype=torch.uint8, device=device))
    local.copy_(torch.tensor([bits, ad, nbytes], dtype=torch.int64))
    try:
        dist.all_gather_into_tensor(gathered, local)  # the one collective of the path
    except (RuntimeError, NotImplementedError):  # backends without the flat form (older gloo)
        parts = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(parts, local)
        gathered.copy_(torch.cat(parts))
    infos = [tuple(int(v) for v in row) for row in gathered.view(world, 3).tolist()]
    bases, end_bit, adler = stream_layout(infos)
    nb = engine.finish(bases[rank], first, last, adler)
    total_bytes = (end_bit + 7) // 8 + 4
    out = engine.out
    # payload exchange: every rank's bytes go to rank 0 in one grouped batch of point-to-point operations (all
    # peers in flight together, one hop each); the first byte of a rank travels apart because rank 0 may have
    # to OR it into a byte the previous rank also wrote.  RCCL moves device memory directly; other backends
    # (gloo, used to rehearse the path) only do point-to-point on host tensors, so the bytes are staged.
    staged = out.is_cuda and dist.get_backend() != "nccl"
    if first:
        final[:nb].copy_(out[:nb])
        edges = scratch.setdefault("edges", torch.zeros(max(1, world), dtype=torch.uint8, device=device))
        ops, spans, landed = [], [], []

        def recv_into(view, r):
            if staged:
                host = torch.empty(view.numel(), dtype=torch.uint8)
                landed.append((view, host))
                view = host
            ops.append(dist.P2POp(dist.irecv, view, r))

        for r in range(1, world):
            off = bases[r] // 8
            stop = total_bytes if r == world - 1 else (bases[r + 1] + 7) // 8
            spans.append((r, off))
            recv_into(edges.narrow(0, r, 1), r)
            for lo in range(off + 1, stop, P2P_PIECE):
                recv_into(final[lo:min(stop, lo + P2P_PIECE)], r)
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for view, host in landed:
            view.copy_(host)
        for r, off in spans:  # in rank order: a rank's first byte lands before the next rank's tail is merged
            # one-element views, not integer indexing: the offsets pass 2^31 on multi-GiB streams
            if bases[r] % 8:
                final.narrow(0, off, 1).bitwise_or_(edges.narrow(0, r, 1))  # the byte two neighbouring ranks share
            else:
                final.narrow(0, off, 1).copy_(edges.narrow(0, r, 1))
    else:
        src = out[:nb].cpu() if staged else out
        ops = [dist.P2POp(dist.isend, src[0:1], 0)]
        for lo in range(1, nb, P2P_PIECE):
            ops.append(dist.P2POp(dist.isend, src[lo:min(nb, lo + P2P_PIECE)], 0))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return total_bytes


class HipPieceEngine:
    """zg_shard_open / zg_shard_next on this rank's GPU, zg_shard_merge / zg_shard_seal on rank 0's."""

    def __init__(self, ctx, d_in_ptr: int, halo: int, n: int, strategy, chunk_bytes: int, batch_bytes: int, packed):
        self.ctx, self.d_in, self.halo, self.n = ctx, d_in_ptr, halo, n
        self.strategy, self.chunk_bytes, self.batch_bytes, self.packed = strategy, chunk_bytes, batch_bytes, packed

    def open(self, last_rank: bool) -> None:
        api.shard_open(self.d_in, self.halo, self.n, self.strategy, last_rank, self.packed.data_ptr(),
                       self.packed.numel(), chunk_bytes=self.chunk_bytes, batch_bytes=self.batch_bytes, ctx=self.ctx)

    def next(self):
        off, nb, bits, last, info = api.shard_next(ctx=self.ctx)
        return off, nb, bits, last, (info.adler if last else None)

    def merge(self, src, n_bits: int, final, base_bit: int, first_run: bool) -> None:
        api.shard_merge(src.data_ptr(), n_bits, final.data_ptr(), final.numel(), base_bit, first_run, ctx=self.ctx)

    def seal(self, final, end_bit: int, adler: int) -> int:
        return api.shard_seal(final.data_ptr(), final.numel(), end_bit, adler, ctx=self.ctx)


def stage_bytes(n: int, bound) -> int:
    """Bytes rank 0 needs to stage a run of n input bytes (`bound` = compress_bound for the run's options)."""
    return bound(n) + 8


class SingleRank:
    """Stands in for torch.distributed when one rank makes the whole stream: the gather is a copy, nothing is sent.
    It lets a one-GPU run go through the same shard calls as a multi-GPU one (bench.py measures what they cost)."""

    @staticmethod
    def get_backend():
        return "single"

    @staticmethod
    def all_gather_into_tensor(out, inp):
        out.copy_(inp)


def compress_sharded_overlapped(engine, dist, torch, rank: int, world: int, device, final=None, stages=None,
                                scratch=None):
    """One sharded compress with the exchange beside the kernels.  Returns the stream length in bytes; on rank 0
    `final[:length]` holds the stream.  `engine.packed` is the rank's own packed buffer; `stages[r]` (rank 0 only,
    r >= 1) a 4-byte aligned uint8 tensor of at least stage_bytes(rank r's input bytes) on `device`."""
    first, last_rank = rank == 0, rank == world - 1
    if scratch is None:
        scratch = {}
    packed = engine.packed
    staged = packed.is_cuda and dist.get_backend() not in ("nccl", "single")  # see compress_sharded
    # RCCL gathers device tensors; the backends used to rehearse the path get the few integers in host memory
    meta = "cpu" if staged else device
    local = scratch.setdefault("piece", torch.zeros(5, dtype=torch.int64, device=meta))
    gathered = scratch.setdefault("pieces", torch.zeros(5 * world, dtype=torch.int64, device=meta))
    # A run can end with the library's device-side self-check failed (ZG_ERR_INTERNAL: the context has then switched to
    # its conservative kernels).  The rank says so in its last row, and every rank starts over, once.
    for attempt in range(2):
        engine.open(last_rank)
        done, got, totals = [False] * world, [0] * world, [None] * world
        mine = None  # (bits, adler) once this rank's last piece is out
        failed = False
        while not all(done):
            if mine is None:
                try:
                    off, nb, bits, last, ad = engine.next()  # queues the next pass's kernels, then waits for the oldest pass
                except api.ZgError as e:
                    if e.code != _lib.ERR_INTERNAL or attempt:
                        raise
                    off, nb, bits, last, ad = 0, 0, 0, True, 0
                    failed = True
                row = [nb, bits, ad if last else 0, (2 if failed else 1) if last else 0, engine.n]
                if last:
                    mine = (bits, ad)
            else:
                off, nb, row = 0, 0, [0, mine[0], mine[1], 2 if failed else 1, engine.n]
            local.copy_(torch.tensor(row, dtype=torch.int64))
            try:
                dist.all_gather_into_tensor(gathered, local)
            except (RuntimeError, NotImplementedError):
                parts = [torch.zeros_like(local) for _ in range(world)]
                dist.all_gather(parts, local)
                gathered.copy_(torch.cat(parts))
            rows = gathered.view(world, 5).tolist()
            ops, landed = [], []
            if first:
                for r in range(1, world):
                    nbr = int(rows[r][0])
                    if stages[r].numel() < got[r] + nbr + 8:
                        raise ValueError(f"staging buffer of rank {r} is too small")
                    for lo in range(0, nbr, P2P_PIECE):
                        view = stages[r].narrow(0, got[r] + lo, min(nbr - lo, P2P_PIECE))
                        if staged:
                            host = torch.empty(view.numel(), dtype=torch.uint8)
                            landed.append((view, host))
                            view = host
                        ops.append(dist.P2POp(dist.irecv, view, r))
                    got[r] += nbr
            elif nb:
                src = packed.narrow(0, off, nb)
                if staged:
                    src = src.cpu()
                for lo in range(0, nb, P2P_PIECE):
                    ops.append(dist.P2POp(dist.isend, src.narrow(0, lo, min(nb - lo, P2P_PIECE)), 0))
            if ops:  # the GPU has the next pass queued: it computes while these bytes travel
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            for view, host in landed:
                view.copy_(host)
            for r in range(world):
                if rows[r][3]:
                    done[r], totals[r] = True, (int(rows[r][1]), int(rows[r][2]), int(rows[r][4]))
            if any(int(rows[r][3]) == 2 for r in range(world)):
                break  # some rank has to start over: so does everybody (runs still open are dropped by the next open)
        else:
            break  # every rank's last piece is in
        if packed.is_cuda:
            torch.cuda.current_stream(device).synchronize()  # nothing of this attempt is in flight when the next begins
    bases, end_bit, adler = stream_layout(totals)
    total_bytes = (end_bit + 7) // 8 + 4
    if packed.is_cuda:
        # RCCL's wait() orders streams, not the host: the received bytes must be in place before the merge (another
        # stream) reads them, and a sender's bytes must have left before its next run overwrites the packed buffer
        torch.cuda.current_stream(device).synchronize()
    if first:
        engine.merge(packed, totals[0][0], final, bases[0], True)
        for r in range(1, world):  # in rank order: a run's first word is ORed into the word the run before ended in
            engine.merge(stages[r], totals[r][0], final, bases[r], False)
        nb = engine.seal(final, end_bit, adler)
        assert nb == total_bytes, (nb, total_bytes)
    return total_bytes
