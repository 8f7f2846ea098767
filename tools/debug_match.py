#!/usr/bin/env python3
"""Counters of k_match summed over its workgroups (library built with -DZG_MATCH_DEBUG)."""
import ctypes as C, os, sys
os.environ["ZLIB_AMD_WALK"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zlib_amd as za
from zlib_amd import _lib
kind = sys.argv[1] if len(sys.argv) > 1 else "text"
n = 64 << 20
if kind == "recbatch":  # records of 4 KiB, each a stream of its own (the small match geometry)
    data = za.corpus("records", n, seed=0x5EED0005)
    za.compress_batch([data[i:i + 4096] for i in range(0, n, 4096)], za.Strategy.Dynamic)
else:
    data = za.corpus(kind, n, seed=0x5EED0004)
    za.compress(data, za.Strategy.Dynamic)
buf = (C.c_ulonglong * 32)()
_lib.lib().zg_match_debug_counters(buf)
v = list(buf)
print(f"{kind}: searches {v[4]} ({v[4] / n:.3f} per position), head steps per search {v[0] / max(1, v[4]):.1f}, measured candidates per search {v[2] / max(1, v[4]):.2f}, chain changes {v[9]}")
print(f"wave trips {v[10]}: head-step slots used {v[0] / max(1, v[10] * 64 * 12):.2f}; the measure block ran in {v[11] / max(1, v[10]):.2f} of them, the retire/claim block in {v[12] / max(1, v[10]):.2f} ({v[13] / max(1, v[12]):.1f} lanes)")
print(f"trips after the tile's last position was claimed: {v[14] / max(1, v[10]):.3f} of all ({v[16] / max(1, v[14]):.1f} walking lanes); trips with at most 8 walking lanes {v[15] / max(1, v[10]):.3f}, at most 2: {v[17] / max(1, v[10]):.3f}")
wg = max(1, v[24])
print(f"per workgroup (100 MHz ticks): whole {v[20] / wg:.0f}, staging {v[21] / wg:.0f} ({v[21] / max(1, v[20]):.2f}), from the first failed claim to the end {v[22] / wg:.0f} ({v[22] / max(1, v[20]):.2f}), first wave waiting for the last {v[23] / wg:.0f}")
print(f"walks finished by the whole wave: {v[18]}")
