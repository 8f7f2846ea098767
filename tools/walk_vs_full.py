#!/usr/bin/env python3
"""Front-end kernel time of the two searches (every position / along the parse) on a few inputs that make long
hash chains: ZLIB_AMD_WALK=0 and =1 in two child processes each, kernel milliseconds from the library's own
profile.  usage: tools/walk_vs_full.py [MiB]"""
import json, os, random, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def inputs(n):
    import zlib_amd as za
    rng = random.Random(5)
    block = bytes(rng.randrange(256) for _ in range(1000))
    line = b"2026-11-11T12:00:00Z INFO request served path=/api/v1/items/%d status=200 bytes=%d\n"
    log = b"".join(line % (rng.randrange(10 ** 6), rng.randrange(10 ** 5)) for _ in range(n // 80 + 1))[:n]
    return {
        "zeros": bytes(n),
        "period 37": (bytes(rng.randrange(256) for _ in range(37)) * (n // 37 + 1))[:n],
        "1000-byte block repeated": (block * (n // 1000 + 1))[:n],
        "log lines": log,
        "structured binary": za.corpus("binary", n, seed=3),
        "text": za.corpus("text", n, seed=3),
    }


def child(mib):
    import torch
    torch.zeros(1, device="cuda")
    import zlib_amd as za
    ctx = za.Context(0)
    out = {}
    for name, data in inputs(mib << 20).items():
        za.compress(data, za.Strategy.Dynamic, ctx=ctx)
        ctx.profile(True)
        ctx.profile_reset()
        z = za.compress(data, za.Strategy.Dynamic, ctx=ctx)
        prof = ctx.profile_get()
        ctx.profile(False)
        out[name] = {"front_ms": sum(prof[k][0] for k in ("k_index", "k_walk", "k_match", "k_match_lo")),
                     "bytes": len(z)}
    print(json.dumps(out))


if __name__ == "__main__":
    mib = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "--child" else 32
    if "--child" in sys.argv:
        child(int(sys.argv[-1]))
    else:
        res = {}
        for mode in ("0", "1", "auto"):
            env = dict(os.environ, ZLIB_AMD_WALK=mode)
            r = subprocess.run([sys.executable, __file__, "--child", str(mib)], env=env, capture_output=True, text=True, timeout=600)
            if r.returncode:
                print(r.stderr[-2000:])
                sys.exit(1)
            res[mode] = json.loads(r.stdout.strip().splitlines()[-1])
        for name in res["0"]:
            a, b, c = (res[m][name] for m in ("0", "1", "auto"))
            assert a["bytes"] == b["bytes"] == c["bytes"], name
            print(f"{name:28s} every position {a['front_ms']:8.2f} ms   along the parse {b['front_ms']:8.2f} ms   by the probe {c['front_ms']:8.2f} ms   ({mib} MiB)")
