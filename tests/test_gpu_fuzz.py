"""A short run of the randomised parity tool (tools/fuzz_parity.py): generated inputs of many shapes, units and
streams of the HIP path against the oracle.  Longer runs: `python tools/fuzz_parity.py SECONDS SEED`."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def _tool():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py")
    spec = importlib.util.spec_from_file_location("fuzz_parity", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_generated_inputs_match_the_oracle():
    ok, cases = _tool().run(0.0, 20261111, max_cases=400)  # a fixed number of cases: the same ones on every box
    assert ok and cases == 400


def test_generated_inputs_match_the_oracle_with_the_search_along_the_parse(monkeypatch):
    """The same kind of run on a context that sends EVERY chunk to k_walk (by default only chunks with long
    chains go there, which generated inputs of this size rarely are)."""
    import zlib_amd as za
    monkeypatch.setenv("ZLIB_AMD_WALK", "1")
    ctx = za.Context(0)
    try:
        ok, cases = _tool().run(0.0, 20261112, max_cases=400, ctx=ctx)
        assert ok and cases == 400
    finally:
        ctx.close()
