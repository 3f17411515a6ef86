#!/usr/bin/env python3
"""
Shim self-check: run the REFERENCE's own unit tests of the gateway
(/root/reference/test/test_aggregate_time.py, 13 tests pinning aggregate_time / per_unit /
deprecation semantics) against the reference's convert.py executing under refshim's xarray/dask
stand-in.  Build container only (needs /root/reference).  Exit code = pytest's.
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import refshim  # noqa: E402

refshim.install()
import pytest  # noqa: E402

sys.exit(pytest.main(["-q", "-p", "no:cacheprovider", "--noconftest", "--rootdir", "/tmp", "-c", "/dev/null",
                      f"{refshim.REFERENCE}/test/test_aggregate_time.py"]))
