#!/usr/bin/env python3
"""Cycles per phase of k_match_lo summed over its workgroups (library built with -DZG_DEBUG_PHASES)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zlib_amd as za
from zlib_amd import _lib
kind = sys.argv[1] if len(sys.argv) > 1 else "text"
n = 64 << 20
data = za.corpus(kind, n, seed=0x5EED0004)
za.compress(data, za.Strategy.Dynamic)
buf = (C.c_ulonglong * 64)()
_lib.lib().zg_debug_counters(buf)
v = list(buf)
tiles = max(1, v[0])
names = ["tiles", "stage", "pass1 loop", "pass1 parked", "restage", "pass2 loop", "pass2 parked", "-", "entries", "parked1", "parked2"]
for k, name in enumerate(names):
    print(f"{name:14s} {v[k]:>16d}  per tile {v[k] / tiles:12.1f}")
for base, name in ((16, "pass1"), (24, "pass2")):
    trips, heads, steps, claim, thin = v[base:base + 5]
    print(f"{name}: trips per wave and tile {trips / tiles / 16:.1f}, walking lanes per trip {heads / max(1, trips):.1f}, "
          f"cycles per trip in the steps {steps / max(1, trips):.0f}, after them {claim / max(1, trips):.0f}, "
          f"trips with 1..4 walking lanes {thin / max(1, trips):.2f}")
