#!/usr/bin/env python3
"""Counters of k_walk summed over its workgroups (library built with -DZG_WALK_DEBUG)."""
import ctypes as C, os, sys
os.environ["ZLIB_AMD_WALK"] = "1"  # the context takes the search along the parse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zlib_amd as za
from zlib_amd import _lib
kind = sys.argv[1] if len(sys.argv) > 1 else "text"
n = 64 << 20
data = za.corpus(kind, n, seed=0x5EED0004)
za.compress(data, za.Strategy.Dynamic)
buf = (C.c_ulonglong * 32)()
_lib.lib().zg_walk_debug_counters(buf)
v = list(buf)
tiles = max(1, v[0])
wt = max(1, v[4])
print(f"{kind}: tiles walked {v[0]} (waves: first launch {v[19]}, mending launch {v[20]})")
print(f"cycles per tile: staging {v[1] / tiles:.0f}, walk loop (first thread) {v[2] / tiles:.0f}, wait at the end for the slowest wave {v[3] / tiles:.0f}")
print(f"trips per wave and tile {v[4] / tiles / 16:.1f} (longest {v[17]}), visited positions per tile {v[9] / tiles:.0f}")
print(f"per trip: walking lanes {v[5] / wt:.1f}, level-2 lanes {v[6] / wt:.1f} (chain steps {v[18] / wt:.1f}), trips with <= 8 busy lanes {v[16] / wt:.2f}")
print(f"measure block ran in {v[10] / wt:.2f} of trips, retire/claim block in {v[8] / wt:.2f} with {v[7] / max(1, v[8]):.1f} lanes; window scans {v[15]}")
print(f"cycles per trip: head steps {v[11] / wt:.0f}, measure+switch {v[12] / wt:.0f}, level 2 {v[13] / wt:.0f}, retire/claim {v[14] / wt:.0f}")
