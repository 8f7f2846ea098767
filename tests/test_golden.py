"""The committed golden vectors (tests/golden/streams.json, made by tests/golden/make_golden.py).

CPU part: the oracle still produces them (guards the checker against drift).  GPU part: the HIP path produces
them, through the C ABI, with no oracle in the loop."""
import hashlib
import json
import os
import zlib as sysz

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "streams.json")) as f:
    GOLD = json.load(f)


def _input(za, spec):
    if "hex" in spec:
        return bytes.fromhex(spec["hex"])
    return za.corpus(spec["corpus"], spec["n"], seed=spec["seed"])


def _check_stream(stream, data, e):
    assert len(stream) == e["stream_len"], e["input"]
    assert hashlib.sha256(stream).hexdigest() == e["stream_sha256"], e["input"]
    if "stream_hex" in e:
        assert stream.hex() == e["stream_hex"]
    assert sysz.decompress(stream) == data
    assert sysz.adler32(data) == e["adler32"]


def _units_digest(toks):
    return hashlib.sha256(b"".join(int(t).to_bytes(4, "little") for t in toks)).hexdigest()


def test_survey_vectors_are_in_the_golden_file():
    """Vectors derived in the survey of the reference (not captured from a Rust run): Fixed streams of '', 'a', 'abcd'."""
    by = {(e["input"].get("hex"), e["strategy"]): e for e in GOLD["streams"] if "hex" in e["input"]}
    assert by[("", "Fixed")]["stream_hex"] == "7801030000000001"
    assert by[(b"a".hex(), "Fixed")]["stream_hex"] == "78014b040000620062"
    assert by[(b"abcd".hex(), "Fixed")]["stream_hex"] == "78014b4c4a4e010003d8018b"


def test_oracle_reproduces_golden_vectors():
    import zlib_amd as za  # corpus generator only
    from oracle import zref as z
    strategies = {"Dynamic": z.DYNAMIC, "Fixed": z.FIXED, "Raw": z.RAW}
    for e in GOLD["streams"]:
        if e["input"].get("n", 0) > 400_000:
            continue  # the 3 MiB case is covered on the GPU; keep the CPU suite short
        data = _input(za, e["input"])
        _check_stream(z.compress_chunked(data, strategies[e["strategy"]], e["chunk_bytes"], fast=True), data, e)
    for e in GOLD["lz77"][:2]:
        toks = z.lz77_zlib(_input(za, e["input"]), fast=True)
        assert len(toks) == e["n_units"] and _units_digest(toks) == e["units_sha256"]
        assert [int(t) for t in toks[:16]] == e["first_units"]


@pytest.fixture(scope="module")
def za_gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    torch.zeros(1, device="cuda")
    import zlib_amd
    zlib_amd.default_context()
    return zlib_amd


@pytest.mark.gpu
def test_device_streams_equal_golden(za_gpu):
    za = za_gpu
    for e in GOLD["streams"]:
        data = _input(za, e["input"])
        got = za.compress(data, getattr(za.Strategy, e["strategy"]), chunk_bytes=e["chunk_bytes"])
        _check_stream(got, data, e)


@pytest.mark.gpu
def test_device_units_equal_golden(za_gpu):
    za = za_gpu
    for e in GOLD["lz77"]:
        toks = za.zlib_compressor().compress_packed(_input(za, e["input"]))
        assert len(toks) == e["n_units"] and _units_digest(toks) == e["units_sha256"], e["input"]
        assert [int(t) for t in toks[:16]] == e["first_units"]


@pytest.mark.gpu
def test_device_record_batches_equal_golden(za_gpu):
    za = za_gpu
    for e in GOLD["records"]:
        spec = e["input"]
        recs = [za.corpus(spec["corpus"], spec["record_bytes"], seed=spec["seed"], offset=spec["record_bytes"] * i)
                for i in range(spec["n_records"])]
        got = za.compress_batch(recs, getattr(za.Strategy, e["strategy"]))
        blob = b"".join(got)
        assert len(blob) == e["total_len"] and hashlib.sha256(blob).hexdigest() == e["sha256_of_concatenation"]
        assert all(sysz.decompress(g) == r for g, r in zip(got, recs))
