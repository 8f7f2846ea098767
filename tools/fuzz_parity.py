#!/usr/bin/env python3
"""Randomised parity run: LZ77 units and whole streams of the HIP path against the oracle on generated inputs of many
shapes (few-symbol motif soups, periodic data, counters in fixed-size records, runs, mixtures; random window /
length parameters and chunk sizes; record batches).  Usage: tools/fuzz_parity.py SECONDS [SEED]"""
import os, random, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zlib_amd as za
from oracle import zref as z

rng = random.Random(1)


def gen(n):
    kind = rng.randrange(8)
    if kind == 0:  # motif soup over a small alphabet
        alpha = bytes(rng.sample(range(256), rng.randint(2, 5)))
        motifs = [bytes(rng.choice(alpha) for _ in range(rng.randint(1, 12))) for _ in range(rng.randint(2, 20))]
        out = bytearray()
        while len(out) < n:
            out += rng.choice(motifs) * rng.randint(1, 8)
            if rng.random() < 0.03:
                out.append(rng.randrange(256))
        return bytes(out[:n])
    if kind == 1:  # records with a counter and constant fields
        rec = rng.choice([8, 12, 16, 20, 24, 32])
        const = bytes(rng.randrange(256) if rng.random() < 0.5 else 0 for _ in range(rec))
        out = bytearray()
        c = rng.randrange(1 << 16)
        while len(out) < n:
            r = bytearray(const)
            r[0:2] = (c & 0xFFFF).to_bytes(2, "little")
            if rng.random() < 0.5:
                r[rec // 2] = rng.randrange(rng.choice([2, 4, 256]))
            out += r
            c += 1
        return bytes(out[:n])
    if kind == 2:  # runs of varying length
        out = bytearray()
        while len(out) < n:
            out += bytes([rng.randrange(rng.choice([2, 256]))]) * rng.randint(1, rng.choice([4, 40, 400, 4000]))
        return bytes(out[:n])
    if kind == 3:
        return za.corpus("text", n, seed=rng.randrange(1 << 30))
    if kind == 4:
        return za.corpus("binary", n, seed=rng.randrange(1 << 30), offset=rng.randrange(1 << 20) * 16)
    if kind == 5:  # periodic with a few disturbances
        per = bytes(rng.randrange(256) for _ in range(rng.randint(1, 300)))
        out = bytearray((per * (n // len(per) + 1))[:n])
        for _ in range(rng.randint(0, 30)):
            out[rng.randrange(n)] ^= 1 + rng.randrange(255)
        return bytes(out)
    if kind == 6:  # concatenation of slices of earlier kinds, repeated far apart
        a, b = gen(n // 3 + 1), gen(n // 3 + 1)
        return (a + b + a[: n // 3 + 1])[:n]
    return bytes(rng.randrange(rng.choice([2, 3, 16, 256])) for _ in range(n))


def first_diff(a, b):
    m = min(len(a), len(b))
    return next((k for k in range(m) if a[k] != b[k]), m)


def run(budget, seed, max_cases=None, ctx=None):
    """Runs cases for `budget` seconds, or exactly `max_cases` cases when that is given (the same cases on every
    machine), on context `ctx` (None: the default one); (all equal, number of cases)."""
    global rng
    rng = random.Random(seed)
    t_end = time.time() + budget
    cases = 0
    t_note = time.time() + 30
    while (cases < max_cases) if max_cases is not None else (time.time() < t_end):
        if time.time() > t_note:  # a line now and then: a silent GPU run is taken to be hung
            print(f"fuzz: {cases} cases so far", flush=True)
            t_note = time.time() + 30
        n = rng.choice([rng.randint(0, 300), rng.randint(300, 20_000), rng.randint(20_000, 150_000)])
        data = gen(n) if n else b""
        mode = rng.randrange(6)
        if mode == 0:  # units, zlib parameters
            got, want = za.zlib_compressor().compress_packed(data, ctx=ctx), z.lz77_zlib(data, fast=True)
            what = "lz77 zlib"
        elif mode == 1:  # units, other parameters (against the verbatim loop: keep the input short)
            data = data[:12_000]
            w = rng.choice([64, 144, 1024, 2048, 4096, 8192, 32768, rng.randint(2048, 32768)])
            mn = rng.choice([3, 4, 5, 6, 8])
            mx = rng.choice([mn + 1, 10, 32, 100, 258])
            mx = max(mx, mn + 1)
            comp = za.LZ77Compressor.with_window_size(w)
            comp.min_match_length, comp.max_match_length = mn, mx
            got, want = comp.compress_packed(data, ctx=ctx), z.lz77_compress(data, 0, w, mn, mx)
            what = f"lz77 w={w} min={mn} max={mx}"
        elif mode == 2:  # streams, several chunks
            chunk = rng.choice([1024, 4096, 16384, 65536, 1 << 20])
            strat = rng.choice([za.Strategy.Dynamic, za.Strategy.Fixed])
            got = za.compress(data, strat, chunk_bytes=chunk, ctx=ctx)
            want = z.compress_chunked(data, z.DYNAMIC if strat == za.Strategy.Dynamic else z.FIXED, chunk, fast=True)
            what = f"stream chunk={chunk} {strat}"
            assert zlib.decompress(got) == data
        elif mode == 4:  # streams cut into several batches (scratch arrays restart a window behind each batch)
            chunk = rng.choice([1024, 4096, 16384, 65536])
            batch = chunk * rng.randint(1, 4)
            got = za.compress(data, za.Strategy.Dynamic, chunk_bytes=chunk, batch_bytes=batch, ctx=ctx)
            want = z.compress_chunked(data, z.DYNAMIC, chunk, fast=True)
            what = f"stream chunk={chunk} batch={batch}"
        elif mode == 5:  # a few large records: isolated slots longer than the window
            recs, pos = [], 0
            while pos < len(data):
                m = rng.choice([30_000, 40_000, 70_000, 100_000])
                recs.append(data[pos:pos + m])
                pos += m
            got = b"".join(za.compress_batch(recs, za.Strategy.Dynamic, ctx=ctx)) if recs else b""
            want = b"".join(z.compress_chunked(r, z.DYNAMIC, 1 << 20, fast=True) for r in recs)
            what = f"batch of {len(recs)} large records"
        else:  # record batch
            recs, pos = [], 0
            while pos < len(data):
                m = rng.choice([1, 7, 100, 1000, 4096, 5000])
                recs.append(data[pos:pos + m])
                pos += m
            got_l = za.compress_batch(recs, za.Strategy.Dynamic, ctx=ctx) if recs else []
            got = b"".join(got_l)
            want = b"".join(z.compress_chunked(r, z.DYNAMIC, 1 << 20, fast=True) for r in recs)
            what = f"batch of {len(recs)} records"
        if got != want:
            k = first_diff(got, want)
            path = os.path.join(ROOT, "gpurun_out", f"fuzz_fail_{seed}_{cases}.bin")
            os.makedirs(os.path.dirname(path), exist_ok=True)
            open(path, "wb").write(data)
            print(f"MISMATCH case {cases} ({what}, n={len(data)}): lengths {len(got)}/{len(want)}, first difference at {k}; input saved to {path}")
            return False, cases
        cases += 1
    print(f"fuzz: {cases} cases equal (seed {seed})")
    return True, cases


if __name__ == "__main__":
    ok, _ = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    sys.exit(0 if ok else 1)
