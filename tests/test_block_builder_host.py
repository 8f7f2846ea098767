"""The single-lane block builder that k_huffman runs on the GPU
(zlib_amd/csrc/zg_huff_core.h), compiled for the host and checked against the
oracle: code lengths, LSB-first codes, dynamic header bits and block size."""
import ctypes as C
import os
import random
import subprocess

import pytest

from oracle import zref as z

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host", "huff_core_host.cpp")
SO = os.path.join(HERE, "host", "libhuff_core_host.so")


@pytest.fixture(scope="module")
def hc():
    core = os.path.join(HERE, "..", "zlib_amd", "csrc", "zg_huff_core.h")
    if not os.path.exists(SO) or max(os.path.getmtime(SRC), os.path.getmtime(core)) > os.path.getmtime(SO):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", SO, SRC])
    return C.CDLL(SO)


def _freqs(tokens):
    ll, d = [0] * 286, [0] * 30
    for t in tokens:
        if t & z.TOK_MARKER:
            ll[z.get_length_code((t >> 16) & 0x1FF)] += 1
            d[z.get_distance_code(t & 0xFFFF)] += 1
        else:
            ll[t & 0xFF] += 1
    ll[256] = max(1, ll[256])
    return ll, d


def _build(hc, tokens, strategy, bfinal):
    ll, d = _freqs(tokens)
    fl, fd = (C.c_uint32 * 288)(*ll), (C.c_uint32 * 32)(*d)
    tl, td, hdr = (C.c_uint32 * 288)(), (C.c_uint32 * 32)(), (C.c_uint32 * 160)()
    hb, pb = C.c_uint32(), C.c_uint64()
    hc.hc_build_block(fl, fd, strategy, int(bfinal), tl, td, hdr, C.byref(hb), C.byref(pb))
    return list(tl), list(td), bytes(hdr), hb.value, pb.value


def _random_tokens(rng, n, lit_alpha, p_match, len_hi=257, dist_hi=32768):
    toks = []
    for _ in range(n):
        if rng.random() < p_match:
            ln = rng.choice([3, 4, 5, rng.randrange(3, len_hi + 1)])
            toks.append(z.marker(ln, rng.choice([5, rng.randrange(5, dist_hi + 1), 1 << rng.randrange(3, 15)]) if dist_hi > 5 else 7))
        else:
            toks.append(lit_alpha[min(int(rng.expovariate(0.15)), len(lit_alpha) - 1)])
    return toks


def _check(hc, toks, strategy, bfinal):
    tl, td, hdr, hb, pb = _build(hc, toks, strategy, bfinal)
    if strategy == z.DYNAMIC:
        ll_len, ll_code, d_len, d_code = z.chunk_tables(toks)
        assert [t >> 16 for t in tl[:286]] == ll_len
        assert [t & 0xFFFF for t in tl[:286]] == ll_code
        assert [t >> 16 for t in td[:30]] == d_len
        assert [t & 0xFFFF for t in td[:30]] == d_code
    blk, bits = z.write_block(toks, strategy, bfinal)
    assert hb + pb == bits
    nfull, rem = divmod(hb, 8)
    assert hdr[:nfull] == blk[:nfull]
    if rem:
        mask = (1 << rem) - 1
        assert hdr[nfull] & mask == blk[nfull] & mask


def test_length_and_distance_code_arithmetic(hc):
    lb, le = z.table("length_base"), z.table("length_extra_bits")
    db, de = z.table("distance_base"), z.table("distance_extra_bits")
    out = (C.c_uint32 * 6)()
    for ln in range(3, 259):
        hc.hc_codes(ln, 1, out)
        code = z.get_length_code(ln)
        assert (out[0], out[1], out[2]) == (code, le[code - 257], lb[code - 257])
    for dist in list(range(1, 3000)) + list(range(3000, 32769, 7)) + [32768, 24577, 24576, 16385, 16384]:
        hc.hc_codes(3, dist, out)
        code = z.get_distance_code(dist)
        assert (out[3], out[4], out[5]) == (code, de[code], db[code])


@pytest.mark.parametrize("strategy", [z.DYNAMIC, z.FIXED])
def test_block_builder_matches_oracle(hc, strategy):
    rng = random.Random(99 + strategy)
    alphas = [list(range(256)), list(b"etaoinshrdlu \n"), [0], [65, 66], list(range(0, 256, 3))]
    for trial in range(150):
        alpha = rng.choice(alphas)
        n = rng.choice([0, 1, 2, 10, 100, 1000, 5000])
        p = rng.choice([0.0, 0.02, 0.2, 0.6, 1.0])
        toks = _random_tokens(rng, n, alpha, p)
        _check(hc, toks, strategy, rng.random() < 0.5)


def test_block_builder_out_of_reference_domain(hc):
    # lone distance symbol, empty block, Fibonacci (depth > 15) histograms
    _check(hc, [97] * 10 + [z.marker(3, 7)], z.DYNAMIC, True)
    _check(hc, [], z.DYNAMIC, True)
    _check(hc, [z.marker(3, 9)] * 5, z.DYNAMIC, False)
    fib = [1, 1]
    while len(fib) < 32:
        fib.append(fib[-1] + fib[-2])
    toks = []
    for s, c in enumerate(fib[1:27]):  # with the end-of-block symbol's 1 an exact Fibonacci series: depth 26
        toks += [s] * c
    _check(hc, toks, z.DYNAMIC, True)
    assert max(z.chunk_tables(toks)[0]) == 15
    # deep distance tree as well
    toks = [1, 2, 3]
    dists = [5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145]
    for i, dd in enumerate(dists):
        toks += [z.marker(3 + i, dd)] * fib[i]
    _check(hc, toks, z.DYNAMIC, True)
    # code-length alphabet stress: every length distinct-ish
    rng = random.Random(5)
    for _ in range(30):
        toks = []
        for s in range(256):
            toks += [s] * rng.choice([0, 1, 1, 2, 4, 9, 30, 200, 3000])
        for _ in range(rng.randrange(0, 300)):
            toks.append(z.marker(rng.randrange(3, 258), rng.randrange(5, 32769)))
        _check(hc, toks, z.DYNAMIC, True)
