#!/usr/bin/env python3
"""How the waves of k_index_r fill the machine (library built with -DZG_INDEX_DEBUG): one 256 MiB batch of a corpus,
begin and end of every wave.  usage: tools/debug_index.py [kind] [MiB]"""
import ctypes as C, os, sys
os.environ["ZLIB_AMD_WALK"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import zlib_amd as za
from zlib_amd import _lib
kind = sys.argv[1] if len(sys.argv) > 1 else "mixed"
n = (int(sys.argv[2]) if len(sys.argv) > 2 else 256) << 20
L = _lib.lib()
data = za.corpus(kind, n, seed=0x5EED0004)
d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
cap = za.compress_bound(n)
d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
za.compress_device(d_in.data_ptr(), n, d_out.data_ptr(), cap)  # warm-up
L.zg_index_debug_reset()
za.compress_device(d_in.data_ptr(), n, d_out.data_ptr(), cap)
buf = (C.c_ulonglong * (1 + 3 * 65536))()
L.zg_index_debug(buf, len(buf))
cnt = min(int(buf[0]), 65536)
rec = [(buf[1 + 3 * i], buf[2 + 3 * i], buf[3 + 3 * i] & 0xFFFFFFFF, buf[3 + 3 * i] >> 32) for i in range(cnt)]
t0 = min(r[0] for r in rec); t1 = max(r[1] for r in rec)
span = (t1 - t0) / 100.0  # microseconds
busy = sum(r[1] - r[0] for r in rec) / 100.0
print(f"{kind}: {cnt} waves, launch spans {span:.0f} us, wave time in all {busy:.0f} us = {busy / span:.0f} waves busy on average")
for key in (5, 4, 3):
    ds = sorted((r[1] - r[0]) / 100.0 for r in rec if r[3] == key)
    if ds:
        print(f"  key {key}: {len(ds)} waves, {ds[0]:.0f} / {ds[len(ds) // 2]:.0f} / {ds[-1]:.0f} us (min / median / max), starts from {min(r[0] for r in rec if r[3] == key) / 100.0 - t0 / 100.0:.0f} us")
# waves running over time, in twentieths of the span
ev = sorted([(r[0], 1) for r in rec] + [(r[1], -1) for r in rec])
run, last, acc = 0, t0, [0.0] * 20
for t, d in ev:
    a, b = last, t
    while a < b:
        k = min(19, int((a - t0) * 20 / max(1, t1 - t0)))
        edge = min(b, t0 + (k + 1) * (t1 - t0) // 20 + 1)
        acc[k] += run * (edge - a)
        a = edge
    run += d
    last = t
print("  waves running, by twentieth of the span:", [round(x / max(1, (t1 - t0) / 20)) for x in acc])
# by kind of MiB in the mixed corpus: segment -> position
ph = (C.c_ulonglong * 8)()
if hasattr(L, "zg_index_debug_phases"):
    L.zg_index_debug_phases(ph)
    v = list(ph)
    gi, gl = max(1, v[7]), max(1, v[6])
    print(f"  waves per CU by the occupancy query: {L.zg_index_waves_per_cu()}")
    print(f"  insert-only groups ({v[7]}): issue {v[0] / gi:.0f}, wait {v[1] / gi:.0f}, results and stores {v[2] / gi:.0f} cycles each")
    print(f"  interleaved groups ({v[6]}): issue {v[3] / gl:.0f}, wait {v[4] / gl:.0f}, results and stores {v[5] / gl:.0f} cycles each")
