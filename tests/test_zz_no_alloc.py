"""zg_adler32_device_noalloc (the entry point that allocates nothing) against zg_adler32_device, through the
C ABI: two calls in a row on one context, the second with a smaller input, plus the limits."""
import zlib as sysz

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    torch.zeros(1, device="cuda")  # bring torch's HIP context up before the library's
    import zlib_amd
    return torch, zlib_amd, zlib_amd.Context(0)


def _both(torch, za, ctx, d_in, n, d_out):
    d_out.fill_(0xDEADBEEF - (1 << 32))
    torch.cuda.synchronize()
    za.adler32_device_noalloc(d_in.data_ptr(), n, d_out.data_ptr(), ctx=ctx)
    got = int(d_out.cpu().item()) & 0xFFFFFFFF
    want = za.adler32_device(d_in.data_ptr(), n, ctx=ctx)
    return got, want


def test_noalloc_matches_the_allocating_entry_point_twice(env):
    torch, za, ctx = env
    big = za.corpus("mixed", 5 * (1 << 20) + 12345, seed=0x5EED0004)
    d_in = torch.frombuffer(bytearray(big), dtype=torch.uint8).cuda()
    d_out = torch.zeros(1, dtype=torch.int32, device="cuda")
    # first call: several segments and both stages of the fold
    got, want = _both(torch, za, ctx, d_in, len(big), d_out)
    assert got == want == sysz.adler32(big)
    # second call, smaller: the scratch still holds the first call's segment values
    small = 300_001
    got, want = _both(torch, za, ctx, d_in, small, d_out)
    assert got == want == sysz.adler32(big[:small])
    # and smaller again, down to nothing
    for n in (4097, 1, 0):
        got, want = _both(torch, za, ctx, d_in, n, d_out)
        assert got == want == sysz.adler32(big[:n])


def test_noalloc_limits(env):
    torch, za, ctx = env
    from zlib_amd import _lib
    d = torch.zeros(16, dtype=torch.uint8, device="cuda")
    d_out = torch.zeros(1, dtype=torch.int32, device="cuda")
    L = _lib.lib()
    # beyond the fixed scratch: refused before anything is launched (the pointer is never read)
    assert L.zg_adler32_device_noalloc(ctx.handle, d.data_ptr(), (4 << 30) + 1, d_out.data_ptr()) == _lib.ERR_CAPACITY
    # no context, no output
    assert L.zg_adler32_device_noalloc(None, d.data_ptr(), 16, d_out.data_ptr()) == _lib.ERR_ARG
    assert L.zg_adler32_device_noalloc(ctx.handle, d.data_ptr(), 16, None) == _lib.ERR_ARG
