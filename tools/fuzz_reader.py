#!/usr/bin/env python3
"""Long corruption fuzz of the NetCDF-4 / HDF5 reader and the DEFLATE decoder (no GPU): truncations, bit flips, byte
splices and random overwrites of every fixture file under tests/golden/nc; a corrupted file must give an error or still
valid data, never a crash.  Run it against the sanitizer build:

    make -C atlite_amd/csrc asan
    LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so) \\
      ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 ATLITE_HIP_LIB=$PWD/atlite_amd/lib/libatlite_hip_asan.so \\
      python tools/fuzz_reader.py [iterations] [seed]
"""
import glob
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from atlite_amd import io  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    files = sorted(glob.glob(str(ROOT / "tests" / "golden" / "nc" / "*.nc")))
    raws = [open(f, "rb").read() for f in files]
    opened = errors = reads = 0
    with tempfile.TemporaryDirectory() as tmp:
        q = os.path.join(tmp, "bad.nc")
        for k in range(n):
            raw = raws[int(rng.integers(len(raws)))]
            b = bytearray(raw)
            kind = int(rng.integers(5))
            if kind == 0:
                b = b[: int(rng.integers(16, len(raw)))]
            elif kind == 1:
                for pos in rng.integers(0, len(raw), size=int(rng.integers(1, 12))):
                    b[int(pos)] ^= 1 << int(rng.integers(8))
            elif kind == 2:  # overwrite a run with random bytes
                a = int(rng.integers(0, len(raw) - 1))
                m = int(rng.integers(1, 64))
                b[a:a + m] = bytes(rng.integers(0, 256, size=min(m, len(raw) - a), dtype=np.uint8))
            elif kind == 3:  # splice: copy one region over another (plausible but wrong offsets / lengths)
                a, c = int(rng.integers(0, len(raw) - 64)), int(rng.integers(0, len(raw) - 64))
                m = int(rng.integers(4, 64))
                b[a:a + m] = raw[c:c + m]
            else:  # damage inside the first 4 KiB (superblock, root group, object headers)
                for pos in rng.integers(0, min(len(raw), 4096), size=int(rng.integers(1, 6))):
                    b[int(pos)] = int(rng.integers(0, 256))
            with open(q, "wb") as f:
                f.write(bytes(b))
            try:
                g = io.NcFile(q)
                opened += 1
                for name, var in list(g.variables.items())[:8]:
                    if var.dtype and 1 <= var.ndim <= 3 and np.prod(var.shape) < 10**6:
                        try:
                            g.read(name)
                            reads += 1
                        except (ValueError, NotImplementedError, MemoryError):
                            errors += 1
                g.close()
            except (ValueError, NotImplementedError, MemoryError):
                errors += 1
    print(f"{n} corrupted files from {len(files)} fixtures: {opened} still opened, {reads} variables read, {errors} clean errors, no crash")


if __name__ == "__main__":
    main()
