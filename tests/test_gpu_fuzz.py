"""GPU: short runs of the randomised differential fuzzers (tests/fuzz_pv_options.py, tests/fuzz_gateway.py)
- random points of the option / aggregation space with hostile input values, GPU result against the
NumPy oracle at the repository tolerance (rtol 1e-10, atol 1e-12 max).  Longer runs: python tests/fuzz_*.py <n> <seed>."""
import importlib.util
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
TOOLS = Path(__file__).resolve().parent


def _run(name, n, seed, monkeypatch):
    spec = importlib.util.spec_from_file_location(name, TOOLS / f"{name}.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, "argv", [name, str(n), str(seed)])
    monkeypatch.setenv("ATLITE_HIP_STREAM", "auto")  # the gateway fuzzer sets it per case; restore afterwards
    return mod.main()


@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_pv_options(monkeypatch, seed):
    assert _run("fuzz_pv_options", 40, seed, monkeypatch) == 0


@pytest.mark.parametrize("seed", [21, 22])
def test_fuzz_gateway(monkeypatch, seed):
    assert _run("fuzz_gateway", 60, seed, monkeypatch) == 0
