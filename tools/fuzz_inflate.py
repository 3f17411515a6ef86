#!/usr/bin/env python3
"""Long differential / corruption fuzz of the library's own DEFLATE decoder (atl_inflate.cpp) against zlib, no GPU:
random payloads of many statistics x zlib level / strategy / memLevel / window, valid streams must decode to the same
bytes; corrupted streams (byte overwrites, bit flips, truncations, splices) must get zlib's verdict from the product
path (fast decoder, zlib on any doubt) and any verdict - but no crash, no out-of-bounds access - from the fast decoder
alone.  Run it against the sanitizer build like tools/fuzz_reader.py:

    LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so) \\
      ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 ATLITE_HIP_LIB=$PWD/atlite_amd/lib/libatlite_hip_asan.so \\
      python tools/fuzz_inflate.py [iterations] [seed]
"""
import sys
import zlib
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from atlite_amd import _lib  # noqa: E402


def inflate(comp, n, which):
    lib = _lib.load()
    dst = np.zeros(max(n, 1), np.uint8)
    src = np.frombuffer(comp, np.uint8) if len(comp) else np.zeros(1, np.uint8)
    rc = lib.atl_inflate_probe(src.ctypes.data, len(comp), dst.ctypes.data, n, which, None)
    return rc, dst[:n].tobytes()


def payload(rng):
    kind = int(rng.integers(7))
    n = int(rng.integers(0, 60000)) if rng.random() < 0.85 else int(rng.integers(60000, 1500000))
    if kind == 0:
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind == 1:
        return rng.integers(0, int(rng.integers(1, 8)), n, dtype=np.uint8).tobytes()
    if kind == 2:  # packed weather-like data through the shuffle filter
        es = int(rng.choice([2, 4, 8]))
        v = (np.cumsum(rng.standard_normal(max(n // es, 1))) * 100).astype({2: np.int16, 4: np.float32, 8: np.float64}[es])
        return v.view(np.uint8).reshape(-1, es).T.copy().tobytes()
    if kind == 3:
        p = int(rng.integers(1, 300))
        return bytes((i % p) * 7 % 256 for i in range(min(n, 30000)))
    if kind == 4:
        return bytes(n)
    if kind == 5:
        words = [bytes(rng.integers(97, 123, int(rng.integers(1, 12)), dtype=np.uint8)) for _ in range(int(rng.integers(2, 60)))]
        return b" ".join(words[int(i)] for i in rng.integers(0, len(words), max(n // 6, 1)))
    a = rng.integers(0, 256, max(n // 2, 1), dtype=np.uint8).tobytes()
    return a + a[: n - len(a)]  # a long-distance repeat


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    valid = corrupt = agree_ok = 0
    for k in range(n):
        d = payload(rng)
        level = int(rng.choice([0, 1, 3, 6, 9]))
        strat = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_FIXED, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY]))
        co = zlib.compressobj(level, zlib.DEFLATED, int(rng.choice([9, 12, 15])), int(rng.choice([1, 5, 9])), strat)
        comp = co.compress(d) + co.flush()
        rc, out = inflate(comp, len(d), 0)
        assert rc == 0 and out == d, ("valid stream", k, len(d), level, strat, rc)
        rc, out = inflate(comp, len(d), 2)
        assert rc == 0 and out == d, ("valid stream, product path", k)
        rc, out = inflate(comp, len(d), 3)  # the device decoder's serial half (atl_inflate_dev.h) on the host
        assert rc == 0 and out == d, ("valid stream, device decoder emulation", k, len(d), level, strat, rc)
        rc, out = inflate(comp, len(d), 4)  # ... and its segment scheme (block finder, count, chain, decode with markers, resolve)
        assert rc == 0 and out == d, ("valid stream, segment scheme emulation", k, len(d), level, strat, rc)
        valid += 1
        for _ in range(3):
            b = bytearray(comp)
            how = int(rng.integers(4))
            if how == 0 and len(b) > 2:
                b = b[: int(rng.integers(1, len(b)))]
            elif how == 1:
                for pos in rng.integers(0, len(b), size=int(rng.integers(1, 5))):
                    b[int(pos)] ^= 1 << int(rng.integers(8))
            elif how == 2:
                for pos in rng.integers(0, len(b), size=int(rng.integers(1, 4))):
                    b[int(pos)] = int(rng.integers(0, 256))
            elif len(b) > 16:
                a, c, m = int(rng.integers(0, len(b) - 8)), int(rng.integers(0, len(b) - 8)), int(rng.integers(1, 8))
                b[a:a + m] = b[c:c + m]
            b = bytes(b)
            want = len(d) if rng.random() < 0.9 else max(0, len(d) + int(rng.integers(-3, 4)))
            try:
                z = zlib.decompress(b)
                ok = len(z) == want
            except Exception:
                ok = False
            rc, out = inflate(b, want, 2)
            assert (rc == 0) == ok, ("verdict differs from zlib", k, how, rc, ok)
            if ok:
                assert out == z
                agree_ok += 1
            inflate(b, want, 0)  # the fast decoder alone: any verdict
            rc3, out3 = inflate(b, want, 3)  # the device decoder's emulation: may refuse, must never accept what zlib rejects
            assert rc3 != 0 or (ok and out3 == z), ("device decoder accepted a stream zlib rejects", k, how)
            rc4, out4 = inflate(b, want, 4)
            assert rc4 != 0 or (ok and out4 == z), ("segment scheme accepted a stream zlib rejects", k, how)
            corrupt += 1
    print(f"{valid} valid streams decoded identically, {corrupt} corrupted streams ({agree_ok} still valid) with zlib's verdict, no crash")


if __name__ == "__main__":
    main()
