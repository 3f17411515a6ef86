#!/usr/bin/env python3
"""Long fuzz of the aggregation-plan builder on the host (atl_agg_check_host: builds the plan and verifies it entry by
entry against its matrix), no GPU: valid random CSR matrices over every tile shape must give consistent plans; hostile
CSR structures (column indices outside the matrix, negative or non-monotonic row pointers, row pointers past the data,
a row length that does not divide the columns) must be refused - never crash.  Run against the sanitizer build like
tools/fuzz_reader.py."""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np
import scipy.sparse as sp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from atlite_amd import _lib  # noqa: E402


def check(N, S, row_len, indptr, indices, data):
    P, dense, err = C.c_int64(), C.c_int64(), C.c_int64()
    rc = _lib.load().atl_agg_check_host(N, S, row_len, indptr.ctypes.data, indices.ctypes.data if len(indices) else None,
                                        data.ctypes.data if len(data) else None, C.byref(P), C.byref(dense), C.byref(err))
    return rc, err.value


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    ok = refused = 0
    for k in range(n):
        tile = rng.choice(["", "16x8", "32x4", "64x2", "flat"])
        if tile:
            os.environ["ATLITE_HIP_TILE"] = str(tile)
        else:
            os.environ.pop("ATLITE_HIP_TILE", None)
        if rng.random() < 0.3:
            os.environ["ATLITE_HIP_FORCE_MFMA"] = "1"
        else:
            os.environ.pop("ATLITE_HIP_FORCE_MFMA", None)
        Y, X, N = int(rng.integers(1, 50)), int(rng.integers(1, 90)), int(rng.integers(0, 70))
        S = Y * X
        M = sp.random(N, S, density=float(rng.choice([0.0, 0.01, 0.2, 1.0])), random_state=int(rng.integers(1 << 30)), format="csr")
        indptr, indices = M.indptr.astype(np.int64), M.indices.astype(np.int32)
        data = rng.normal(size=M.nnz)
        row_len = X if rng.random() < 0.7 else 0
        hostile = int(rng.integers(8))
        valid = True
        if hostile == 1 and len(indices):
            indices = indices.copy(); indices[int(rng.integers(len(indices)))] = int(rng.choice([S, S + 7, -1, 2**31 - 1, -2**31])); valid = False
        elif hostile == 2 and N > 0:
            # an INTERIOR pointer (the last one defines how long indices / data are: that is the caller's contract)
            if N > 1:
                indptr = indptr.copy(); indptr[int(rng.integers(1, N))] += int(rng.choice([5, 10**6, 2**40])); valid = None  # may or may not break monotony
        elif hostile == 3 and N > 1:
            indptr = indptr.copy(); indptr[int(rng.integers(1, N))] = -3; valid = False
        elif hostile == 4:
            row_len = int(rng.choice([S + 1, 7 if S % 7 else S + 3, -4])); valid = S > 0 and row_len > 0 and S % row_len == 0
        elif hostile == 5 and N > 0:
            indptr = indptr.copy(); indptr[0] = int(rng.choice([1, -1])); valid = False
        rc, err = check(N, S, row_len, np.ascontiguousarray(indptr), np.ascontiguousarray(indices), data)
        if valid is True:
            assert rc == 0 and err == 0, ("valid matrix", k, rc, err, N, S, row_len, tile)
            ok += 1
        elif valid is False:
            assert rc != 0, ("hostile structure accepted", k, hostile, N, S, row_len)
            refused += 1
        else:
            assert rc != 0 or err == 0
    print(f"{n} matrices: {ok} consistent plans, {refused} hostile structures refused, no crash")


if __name__ == "__main__":
    main()
