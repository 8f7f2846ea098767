#!/usr/bin/env python3
"""Benchmark of the Dynamic-strategy compress path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one compress pass over the whole synthetic corpus, input already
resident in HBM, output left in HBM (rank 0 holds the assembled zlib stream).
The default workload is the one the headline metric names: Strategy::Dynamic on
the 8 GiB mixed text/binary corpus, compressed into ONE zlib stream by however
many GPUs take part (scaling "strong": the corpus stays 8 GiB, N GPUs take
8/N GiB each).  The other workloads keep a fixed number of bytes per GPU
(scaling "weak").  The corpus is cut into contiguous runs of 1 MiB chunks, one
run per rank (each rank also sees the 32 KiB of input before its run); the
ranks compress their runs pass by pass; after every pass one small all_gather
over RCCL says how many bytes each rank has finished, and those bytes travel
to rank 0 while the next pass's kernels run; rank 0 merges the runs at their
bit offsets (zlib_amd/sharding.py).  PyTorch is used for device memory, streams
and torch.distributed only.

On one GPU with the default workload the same JSON line carries, under
"other_configs", short runs of the other single-GPU configurations (64 MiB of
text, 4 GiB of stored blocks, 4 GiB of 4 KiB records) and of the headline
workload through the shard calls.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak

STRONG = {"mixed8g"}  # workloads whose total size is fixed; the others have a fixed size per GPU
WORKLOADS = {
    # name: (corpus kind, bytes (in total if STRONG, else per GPU), chunk bytes, seed, strategy)
    "mixed8g": ("mixed", 8 << 30, 1 << 20, 0x5EED0004, "Dynamic"),   # the configuration the headline metric is quoted on
    "text64m": ("text", 64 << 20, 1 << 20, 0x5EED0001, "Dynamic"),   # 64 MiB of text on a single GPU
    "mixed1g": ("mixed", 1 << 30, 1 << 20, 0x5EED0004, "Dynamic"),   # what one of 8 GPUs gets of mixed8g
    "text1g": ("text", 1 << 30, 1 << 20, 0x5EED0001, "Dynamic"),
    # the three kinds of MiB segment the mixed corpus is made of (2 : 1 : 1), each alone
    "text256m": ("text", 256 << 20, 1 << 20, 0x5EED0001, "Dynamic"),
    "random256m": ("random", 256 << 20, 1 << 20, 0x5EED0003, "Dynamic"),
    "binary256m": ("binary", 256 << 20, 1 << 20, 0x5EED0004, "Dynamic"),
    "raw4g": ("random", 4 << 30, 1 << 20, 0x5EED0003, "Raw"),        # configs[2]: stored blocks + Adler-32
    "raw1g": ("random", 1 << 30, 1 << 20, 0x5EED0003, "Raw"),
    # configs[4]: records of 4 KiB, each compressed as a zlib stream of its own ("chunk" = record size)
    "records4g": ("records", 4 << 30, 4096, 0x5EED0005, "Dynamic"),
    "records256m": ("records", 256 << 20, 4096, 0x5EED0005, "Dynamic"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("ZG_BENCH_WORKLOAD", "mixed8g"), choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="the headline workload only")
    return ap.parse_args()


GEN_THREADS = 8
GEN_PIECE = 32 << 20


def upload_corpus(torch, za, kind, seed, offset, n, device):
    """Corpus bytes [offset, offset+n) in device memory, generated on the host by a few threads, piece by piece."""
    from concurrent.futures import ThreadPoolExecutor
    dev = torch.empty(n + 64, dtype=torch.uint8, device=device)
    bufs = [torch.empty(min(n, GEN_PIECE), dtype=torch.uint8).pin_memory() for _ in range(min(GEN_THREADS, -(-n // GEN_PIECE)))]

    def fill(j, done, m):
        za.corpus_into(kind, bufs[j].data_ptr(), m, seed=seed, offset=offset + done)
        return j, done, m

    with ThreadPoolExecutor(len(bufs)) as pool:
        done = 0
        while done < n:
            jobs = []
            for j in range(len(bufs)):
                if done >= n:
                    break
                m = min(GEN_PIECE, n - done)
                jobs.append(pool.submit(fill, j, done, m))
                done += m
            for job in jobs:
                j, at, m = job.result()
                dev[at:at + m].copy_(bufs[j][:m], non_blocking=False)
    return dev


def verify_stream(za, stream, kind, seed, total):
    """Inflates the zlib stream with system zlib piece by piece and compares with the generator (bounded memory)."""
    import zlib as sysz
    from concurrent.futures import ThreadPoolExecutor
    d = sysz.decompressobj()
    view = memoryview(stream)
    pos, feed = 0, 16 << 20

    def expect(off, m):
        return za.corpus(kind, m, seed=seed, offset=off)

    with ThreadPoolExecutor(GEN_THREADS) as pool:
        for at in range(0, len(view), feed):
            out = d.decompress(view[at:at + feed])
            if pos + len(out) > total:
                return False
            part = -(-len(out) // GEN_THREADS) if out else 0
            part = max(part, 4 << 20)
            jobs = [(o, pool.submit(expect, pos + o, min(part, len(out) - o))) for o in range(0, len(out), part)]
            mv = memoryview(out)
            for o, job in jobs:
                want = job.result()
                if mv[o:o + len(want)] != want:
                    return False
            pos += len(out)
        tail = d.flush()
    return pos + len(tail) == total and not tail and d.eof and not d.unused_data


TRAFFIC_FILE = "r3_traffic.json"


def measured_traffic(workload, kernel, live_avg_ms):
    """(HBM bytes per launch of `kernel`, stale?) as collected with tools/collect_profiles.sh -- counters cannot be
    read from inside the timed process.  The profile also holds the kernel's mean duration in its own run: when
    this run's differs by more than a fifth, the kernel has changed since and the figure is marked stale.
    (None, None) when the committed profile has no entry for this workload and kernel."""
    try:
        with open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)) as f:
            t = json.load(f)["workloads"]
        e = t[workload][kernel if kernel in t[workload] else kernel + "_r"]
        then = e.get("avg_ms_in_stats_run")
        stale = None if not then else abs(live_avg_ms / then - 1.0) > 0.2
        return e["hbm_bytes_per_launch"], stale
    except (OSError, KeyError, ValueError):
        return None, None


def cpu_baseline(za, kind, seed, total, chunk, torch=None, ctx=None, device=None):
    """The reference algorithm (oracle, verbatim exhaustive matcher) on this host's cores, on a bounded sample; and
    this library on the SAME slices (each one block with its 32 KiB window), for the compression ratio side by side."""
    from oracle import zref
    cores = min(16, len(os.sched_getaffinity(0)))
    slice_len = 768 * 1024 if kind == "text" else 160 * 1024
    n_chunks = max(1, total // chunk)
    slices, blobs, pos = [], [], 0
    for i in range(cores):
        c = (i * n_chunks) // cores  # spread evenly over the corpus
        if kind == "mixed":
            c = (c // 4) * 4 + (i % 4)  # visit every segment type of the mixed corpus
            c = min(c, n_chunks - 1)
        start = c * chunk
        halo = min(32768, start)
        blob = za.corpus(kind, halo + slice_len, seed=seed, offset=start - halo)
        blobs.append(blob)
        slices.append((pos + halo, slice_len, halo))
        pos += len(blob)
    data = b"".join(blobs)
    t0 = time.perf_counter()
    bits = zref.bench_slices(data, slices, cores)
    dt = time.perf_counter() - t0
    nbytes = slice_len * len(slices)
    same = {}
    if torch is not None:
        gpu_bits = []
        for blob, (_, _, halo) in zip(blobs, slices):
            d = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
            info = za.shard_begin(d.data_ptr(), halo, slice_len, za.Strategy.Dynamic, True, chunk_bytes=slice_len, ctx=ctx)
            gpu_bits.append(int(info.total_bits))
        ref_ratio, own_ratio = sum(bits) / 8 / nbytes, sum(gpu_bits) / 8 / nbytes
        same = {"build_ratio_same_slices": own_ratio, "ratio_delta": own_ratio / ref_ratio - 1.0,
                "slices_with_identical_bit_count": sum(a == b for a, b in zip(bits, gpu_bits))}
    return {
        **same,
        "value": nbytes / dt / 1e9, "unit": "GB/s", "cores": cores, "kind": "port",
        "sample": f"{len(slices)} slices of {slice_len // 1024} KiB (each with its 32 KiB window) spread over the "
                  f"corpus, one thread per slice, {dt:.1f} s wall; reference-style single Dynamic block per slice",
        "sample_ratio": sum(bits) / 8 / nbytes,
    }


def device_round_trip(torch, za, ctx, d_stream, stream_bytes, offsets, chunk, d_in, n, device):
    """Inflates the stream on the device, every block at once (zg_decompress_blocks_device: the library's own inflate,
    which also checks the Adler-32), and compares with the input there.  (ok, milliseconds of the inflate)."""
    back = torch.empty(n + 64, dtype=torch.uint8, device=device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    za.decompress_blocks_device(d_stream.data_ptr(), stream_bytes, offsets, chunk, back.data_ptr(), n, ctx=ctx)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    ok = bool(torch.equal(back[:n], d_in[:n]))
    del back
    return ok, ms


def run_workload(name, steps, warmup, env, verify=True, shard_path=False, exchange="overlapped"):
    """Uploads the workload's corpus, times `steps` passes of the hot path over it and checks the result.  Returns a
    dictionary of what was measured (on every rank; the whole-job figures are valid on rank 0)."""
    torch, dist, za, sharding = env["torch"], env["dist"], env["za"], env["sharding"]
    rank, world, device, ctx = env["rank"], env["world"], env["device"], env["ctx"]
    kind, per_gpu, chunk, seed, strat_name = WORKLOADS[name]
    strong = name in STRONG
    total = per_gpu if strong else per_gpu * world  # bytes of the whole job
    start, n, halo = sharding.plan_shards(total, chunk, world)[rank]
    d_in = upload_corpus(torch, za, kind, seed, start - halo, halo + n, device)
    strategy = getattr(za.Strategy, strat_name)
    if world > 1 and strat_name != "Dynamic":
        raise SystemExit("the sharded path covers Dynamic/Fixed blocks only")
    batch = kind == "records"
    final = d_off = d_out = None
    bound = lambda m: za.compress_bound(m, strategy, chunk_bytes=chunk)
    if batch:
        # records are independent streams: every rank compresses its own records, nothing is exchanged
        n_rec = n // chunk
        cap = za.compress_batch_bound(chunk, n_rec, strategy)
        d_out = torch.empty(cap, dtype=torch.uint8, device=device)
        d_off = torch.zeros(n_rec + 1, dtype=torch.int64, device=device)

        def step():
            return za.compress_batch_device(d_in.data_ptr() + halo, chunk, 0, n_rec, d_out.data_ptr(), cap,
                                            d_off.data_ptr(), strategy, ctx=ctx)
    elif world == 1 and not shard_path:
        cap = bound(n)
        d_out = torch.empty(cap, dtype=torch.uint8, device=device)

        def step():
            return za.compress_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, strategy, chunk_bytes=chunk, ctx=ctx)
    elif exchange == "overlapped":
        # every rank's run in passes of SHARD_PASS bytes; the bytes of pass k travel while pass k + 1 is computed
        packed = torch.empty(bound(n) + 8, dtype=torch.uint8, device=device)
        stages = None
        if rank == 0:
            final = torch.empty(bound(total) + 8, dtype=torch.uint8, device=device)
            plan = sharding.plan_shards(total, chunk, world)
            stages = [None] + [torch.empty(sharding.stage_bytes(plan[r][1], bound), dtype=torch.uint8, device=device)
                               for r in range(1, world)]
        engine = sharding.HipPieceEngine(ctx, d_in.data_ptr(), halo, n, strategy, chunk, SHARD_PASS, packed)
        scratch = {}
        d = dist if world > 1 else sharding.SingleRank

        def step():
            return sharding.compress_sharded_overlapped(engine, d, torch, rank, world, device, final=final,
                                                        stages=stages, scratch=scratch)
    else:
        d_out = torch.empty(bound(n) + 8, dtype=torch.uint8, device=device)
        if rank == 0:
            final = torch.empty(bound(total) + 8, dtype=torch.uint8, device=device)
        engine = sharding.HipEngine(ctx, d_in.data_ptr(), halo, n, strategy, chunk, d_out)
        scratch = {}

        def step():
            return sharding.compress_sharded(engine, dist, torch, rank, world, device, final=final, scratch=scratch)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(warmup):
        out_n = step()
    # One untimed step with HIP events around every kernel launch: the breakdown per kernel, and which kernel
    # dominates.  In the timed steps only that kernel's launches carry events (they give the roofline's duration);
    # events around all of them cost a step about one per cent.  ZG_BENCH_TIMERS=all / 0: every kernel / none in
    # the timed steps (measurement hook).
    timers = os.environ.get("ZG_BENCH_TIMERS", "dominant")
    ctx.profile(True)
    ctx.profile_reset()
    sync()
    out_n = step()
    sync()
    prof = ctx.profile_get()
    dominant = max(((k, v) for k, v in prof.items() if k != "misc"), key=lambda kv: kv[1][0])[0]
    ctx.profile(timers != "0", only=None if timers == "all" else dominant)
    ctx.profile_reset()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        out_n = step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof_timed = ctx.profile_get()
    ctx.profile(False)

    verified, inflate_ms = None, None
    if batch:
        if world > 1:  # whole-job output size
            t = torch.tensor([out_n], dtype=torch.int64, device=device)
            dist.all_reduce(t)
            out_n_total = int(t.item())
        else:
            out_n_total = out_n
        if rank == 0 and verify:
            import zlib as sysz
            off = d_off.cpu().tolist()
            verified = off[-1] == out_n
            for i in range(0, n_rec, max(1, n_rec // 2000)):  # a spread sample of this rank's records
                got = sysz.decompress(d_out[off[i]:off[i + 1]].cpu().numpy().tobytes())
                verified = verified and got == za.corpus(kind, chunk, seed=seed, offset=start + i * chunk)
            if not verified:
                raise SystemExit("record round trip through system zlib FAILED")
        out_n = out_n_total
    elif rank == 0 and verify:
        d_stream = final if final is not None else d_out
        if world == 1 and strat_name != "Raw":
            offsets = za.last_block_offsets(ctx=ctx)
            if shard_path:
                offsets = [o + sharding.ZLIB_HEADER_BITS for o in offsets]  # a run's offsets count from its first block
            ok, inflate_ms = device_round_trip(torch, za, ctx, d_stream, out_n, offsets, chunk, d_in[halo:], n, device)
            if not ok:
                raise SystemExit("round trip through the library's inflate FAILED")
        if verify == "full":
            stream = d_stream[:out_n].cpu().numpy()
            verified = verify_stream(za, stream, kind, seed, total)
            del stream
            if not verified:
                raise SystemExit("round trip through system zlib FAILED")
    res = {"name": name, "kind": kind, "strategy": strat_name, "chunk": chunk, "seed": seed, "total": total, "n": n,
           "strong": strong, "batch": batch, "dt": dt, "steps": steps, "out_n": out_n, "prof": prof,
           "dominant": dominant, "dominant_timed": prof_timed[dominant],
           "verified": verified, "inflate_ms": inflate_ms}
    del d_in, d_out, final
    torch.cuda.empty_cache()
    return res


def kernel_roofline(res, workload_for_traffic, world):
    """The dominant kernel of a step by event time, priced on algorithmic bytes; its duration is the mean over its
    launches in the timed steps."""
    steps = res["steps"]
    kname, (kms, kn) = res["dominant"], res["dominant_timed"]
    k_avg_ms = kms / max(1, kn)
    launches_per_step = kn / steps
    ratio = res["out_n"] / res["total"]
    bytes_per_launch = res["n"] * (1.0 + ratio) / max(1.0, launches_per_step)  # this rank's share
    if k_avg_ms <= 0:  # ZG_BENCH_TIMERS=0: nothing was timed per kernel
        return {"bound": "hbm", "kernel": kname, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                "traffic": None, "kernel_avg_ms": None, "launches_per_step": None}
    achieved = bytes_per_launch / (k_avg_ms * 1e-3) / 1e9
    traffic, stale = measured_traffic(workload_for_traffic, kname, k_avg_ms) if world == 1 else (None, None)
    return {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_stale": stale,
            "kernel_avg_ms": k_avg_ms, "launches_per_step": launches_per_step,
            "algorithmic_bytes_per_launch": bytes_per_launch}


SHARD_PASS = 128 << 20  # input bytes per pass of a rank's run when the exchange runs beside the kernels
OTHER_CONFIGS = ("text64m", "raw4g", "records4g")  # the other single-GPU configurations, measured after the headline


def main():
    args = parse_args()
    if os.environ.get("ZG_BENCH_HANG_DUMP"):  # seconds after which every thread's stack is printed and the run ends
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["ZG_BENCH_HANG_DUMP"]), exit=True)
    import torch
    import torch.distributed as dist

    import zlib_amd as za
    from zlib_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    # ZG_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks share cards)
    backend = os.environ.get("ZG_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    ctx = za.Context(dev_index)
    env = {"torch": torch, "dist": dist, "za": za, "sharding": sharding, "rank": rank, "world": world,
           "device": device, "ctx": ctx}
    exchange = os.environ.get("ZG_BENCH_EXCHANGE", "overlapped")  # "serial": exchange after all kernels (compress_sharded)

    res = run_workload(args.workload, args.steps, args.warmup, env, verify=False if args.no_verify else "full",
                       exchange=exchange)
    kind, chunk, seed, strat_name = res["kind"], res["chunk"], res["seed"], res["strategy"]
    total, batch, strong, out_n = res["total"], res["batch"], res["strong"], res["out_n"]
    per_gpu = total // world

    if rank == 0:
        ms_step = res["dt"] / args.steps * 1e3
        value = total * args.steps / res["dt"] / 1e9
        ratio = out_n / total
        roof = kernel_roofline(res, args.workload, world)
        roof["traffic_note"] = ("HBM bytes per launch from FETCH_SIZE / WRITE_SIZE (separate rocprofv3 --pmc passes over "
                                f"this command, profiles/{TRAFFIC_FILE}); null if not collected for this workload; "
                                "stale: the kernel's duration has moved by more than a fifth since the counters were taken")
        roof["note"] = "algorithmic bytes = input bytes read once + compressed bytes written once"
        if world == 1:
            par = "single GPU"
        elif batch:
            par = "records sharded over ranks, no exchange"
        elif exchange == "overlapped":
            par = (f"chunks sharded over ranks; per pass of {SHARD_PASS >> 20} MiB one all_gather of piece sizes and the "
                   "pass's bytes sent to rank 0 beside the next pass's kernels; rank 0 merges the runs at their bit offsets")
        else:
            par = "chunks sharded over ranks; all_gather of bit totals + payload send to rank 0"
        line = {
            "metric": f"compress input GB/s ({strat_name})", "value": value, "unit": "GB/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": (f"Strategy::{strat_name} on {total // chunk} synthetic records of {chunk} B "
                                    f"({per_gpu >> 20} MiB per GPU), one zlib stream per record, {world} MI355X") if batch else
                                   (f"Strategy::{strat_name} on {total >> 20} MiB synthetic {kind} ({per_gpu >> 20} MiB per GPU), "
                                    f"{chunk >> 10} KiB chunks with 32 KiB window, {world} MI355X"),
                       "name": args.workload, "chunk_bytes": chunk,
                       "total_bytes": total, "compressed_bytes": out_n, "ratio": ratio,
                       "round_trip_system_zlib": res["verified"],
                       "round_trip_on_device": None if res["inflate_ms"] is None else True,
                       "inflate_on_device_ms": res["inflate_ms"],
                       "parallelism": par},
            "roofline": roof,
            "kernels_ms_per_step": {k: v[0] for k, v in res["prof"].items() if v[1]},
            "kernels_ms_per_step_note": "HIP events around every launch in one untimed step before the timed ones; in "
                                        "the timed steps only the dominant kernel's launches carry events (roofline)",
        }
    headline = args.workload == "mixed8g"
    if world == 1 and headline and not args.no_others:
        # the other single-GPU configurations and the shard calls on one rank, a few passes each
        others = {}
        for name in OTHER_CONFIGS:
            r = run_workload(name, 3, 1, env, verify=not args.no_verify and ("full" if name != "records4g" else True))
            roof = kernel_roofline(r, name, world)
            others[name] = {"metric": f"compress input GB/s ({r['strategy']})", "value": r["total"] * r["steps"] / r["dt"] / 1e9,
                            "unit": "GB/s", "ms_per_step": r["dt"] / r["steps"] * 1e3, "steps": r["steps"], "warmup": 1,
                            "total_bytes": r["total"], "chunk_bytes": r["chunk"], "ratio": r["out_n"] / r["total"],
                            "round_trip_system_zlib": r["verified"], "inflate_on_device_ms": r["inflate_ms"],
                            "dominant_kernel": roof["kernel"], "kernel_avg_ms": roof["kernel_avg_ms"],
                            "roofline_frac": roof["frac"], "traffic": roof["traffic"]}
        r = run_workload(args.workload, 3, 1, env, verify=not args.no_verify, shard_path=True)
        ms = r["dt"] / r["steps"] * 1e3
        others["mixed8g_shard_calls"] = {
            "what": f"the headline workload through zg_shard_open/next/merge/seal on one rank, passes of {SHARD_PASS >> 20} MiB "
                    "(what a rank of a multi-GPU run executes, with nothing to exchange)",
            "value": r["total"] * r["steps"] / r["dt"] / 1e9, "unit": "GB/s", "ms_per_step": ms, "steps": r["steps"],
            "ms_over_compress_device": ms - ms_step, "compressed_bytes_equal": r["out_n"] == out_n,
            "inflate_on_device_ms": r["inflate_ms"]}
        line["other_configs"] = others
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and strat_name == "Dynamic" and not batch:
            line["cpu_baseline"] = cpu_baseline(za, kind, seed, total, chunk, torch, ctx, device)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
