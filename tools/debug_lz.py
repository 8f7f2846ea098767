"""First token where the device matcher and the oracle disagree (debug aid): tools/debug_lz.py KIND N"""
import sys
sys.path.insert(0, ".")
import torch
torch.zeros(1, device="cuda")
import zlib_amd as za
from oracle import zref as z
kind, n = sys.argv[1], int(sys.argv[2])
data = za.corpus(kind, n)
got = za.zlib_compressor().compress_packed(data)
want = z.lz77_zlib(data, fast=True)
print(len(got), len(want))
pos = 0
for i, (a, b) in enumerate(zip(got, want)):
    if a != b:
        print("token", i, "pos", pos, "got", z.tok_str(a), "want", z.tok_str(b), data[pos:pos + 8])
        break
    pos += ((a >> 16) & 0x1FF) if a & 0x80000000 else 1
else:
    print("equal prefix")
