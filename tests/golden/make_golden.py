#!/usr/bin/env python3
"""Writes tests/golden/streams.json: expected outputs of the compress path for seeded inputs.

The reference crate cannot be run in the build image (no Rust toolchain), so these vectors are produced by
the CPU oracle (oracle/zref.cpp), which is itself pinned by the reference's known-answer tests
(tests/test_oracle_kat.py).  Every stream was also inflated with system zlib when the file was made.  The file
holds data only: how to regenerate each input (corpus kind / seed / length, or the literal bytes), and what
the path must produce for it.

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys
import zlib as sysz

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import zref as z  # noqa: E402
import zlib_amd as za  # noqa: E402  (corpus generator only; no device call)

LITERALS = [b"", b"a", b"abcd", b"aaaaa", b"hello hello hello hello", b"a" * 1200,
            b"abcabcabcabcabcabcabcabc_abcabc", bytes(range(256)) * 3]
CORPORA = [("text", 30_000, 0x5EED0001, 1 << 20), ("text", 300_000, 0x5EED0001, 65536),
           ("binary", 200_000, 0x5EED0002, 32768), ("random", 100_000, 0x5EED0003, 16384),
           ("mixed", 3 << 20, 0x5EED0004, 1 << 20), ("records", 40_960, 0x5EED0005, 4096)]
STRATEGIES = {"Dynamic": z.DYNAMIC, "Fixed": z.FIXED, "Raw": z.RAW}


def entry(data, strategy, chunk, src):
    stream = z.compress_chunked(data, STRATEGIES[strategy], chunk, fast=True)
    assert sysz.decompress(stream) == data
    e = {"input": src, "strategy": strategy, "chunk_bytes": chunk, "stream_len": len(stream),
         "stream_sha256": hashlib.sha256(stream).hexdigest(), "adler32": sysz.adler32(data)}
    if len(stream) <= 96:
        e["stream_hex"] = stream.hex()
    return e


def main():
    out = {"streams": [], "lz77": [], "records": []}
    for lit in LITERALS:
        for st in STRATEGIES:
            out["streams"].append(entry(lit, st, 1 << 20, {"hex": lit.hex()}))
    for kind, n, seed, chunk in CORPORA:
        data = za.corpus(kind, n, seed=seed)
        for st in ("Dynamic", "Fixed"):
            out["streams"].append(entry(data, st, chunk, {"corpus": kind, "n": n, "seed": seed}))
    # LZ77 unit streams (zlib parameters), as (count, sha256 of the little-endian packed units)
    for kind, n, seed in (("text", 200_000, 0x5EED0001), ("binary", 100_000, 0x5EED0002), ("mixed", 150_000, 0x5EED0004)):
        toks = z.lz77_zlib(za.corpus(kind, n, seed=seed), fast=True)
        blob = b"".join(int(t).to_bytes(4, "little") for t in toks)
        out["lz77"].append({"input": {"corpus": kind, "n": n, "seed": seed}, "n_units": len(toks),
                            "units_sha256": hashlib.sha256(blob).hexdigest(), "first_units": [int(t) for t in toks[:16]]})
    # record batches: every record its own stream
    recs = [za.corpus("records", 4096, seed=0x5EED0005, offset=4096 * i) for i in range(64)]
    for st in ("Dynamic", "Fixed"):
        streams = [z.compress_chunked(r, STRATEGIES[st], 1 << 20, fast=True) for r in recs]
        out["records"].append({"input": {"corpus": "records", "record_bytes": 4096, "n_records": 64, "seed": 0x5EED0005},
                               "strategy": st, "total_len": sum(map(len, streams)),
                               "sha256_of_concatenation": hashlib.sha256(b"".join(streams)).hexdigest()})
    with open(os.path.join(ROOT, "tests", "golden", "streams.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(len(out["streams"]), "streams,", len(out["lz77"]), "unit streams,", len(out["records"]), "record batches")


if __name__ == "__main__":
    main()
