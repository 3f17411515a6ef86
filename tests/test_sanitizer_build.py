"""
CPU: the host code that parses untrusted bytes (NetCDF-4 / HDF5 reader, DEFLATE decoder), clips user polygons and
builds aggregation plans, rebuilt under AddressSanitizer + UndefinedBehaviorSanitizer (`make asan`) and driven by the
file-reader suite - corruption fuzzing and the differential DEFLATE fuzz included - and the host-logic suite.
A sanitizer report aborts the child process, which fails this test.
"""
import os
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def test_reader_inflate_clipper_and_plan_builder_under_asan_ubsan():
    rt = subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "-print-file-name=libclang_rt.asan-x86_64.so"],
                        capture_output=True, text=True).stdout.strip()
    if not rt or not os.path.exists(rt):
        pytest.skip("no AddressSanitizer runtime in this toolchain")
    env = dict(os.environ, PYTEST_ADDOPTS="")
    r = subprocess.run([str(ROOT / "tools" / "run_asan_tests.sh"), "-x"], capture_output=True, text=True, env=env,
                       timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail


def test_reader_worker_pool_under_tsan():
    """The same reader driven through its worker pool (parallel chunk inflate, eight threads) under ThreadSanitizer
    (`make tsan`): a data-race report aborts the child process."""
    rt = subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "-print-file-name=libclang_rt.tsan-x86_64.so"],
                        capture_output=True, text=True).stdout.strip()
    if not rt or not os.path.exists(rt):
        pytest.skip("no ThreadSanitizer runtime in this toolchain")
    env = dict(os.environ, PYTEST_ADDOPTS="")
    r = subprocess.run([str(ROOT / "tools" / "run_tsan_tests.sh"), "-x", "-k", "not test_c_struct_layout"], capture_output=True, text=True,
                       env=env, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail and "ThreadSanitizer" not in tail, tail
