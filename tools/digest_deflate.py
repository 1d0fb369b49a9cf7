#!/usr/bin/env python3
"""digest_deflate.py [--quick] [--levels=L,...] - compress a fixed, seeded set of inputs at
every level through the library LIBDEFLATE_AMD_LIB points at and print one
sha256 per (level, case): two builds whose kernels make the same choices print
the same lines (`diff` of two runs), which is how a re-scheduling of the
compress kernel is checked to be output-neutral.  Every stream is also decoded
with zlib."""
import hashlib
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from tests import datagen  # noqa: E402
from tests.test_deflate_gpu import _weird_chunk  # noqa: E402

WB = {"deflate": -15, "zlib": 15, "gzip": 31}


def main():
    from libdeflate_amd import api
    quick = "--quick" in sys.argv
    rng = np.random.default_rng(0x0D16E57)
    edges = [0, 1, 3, 52, 53, 4095, 4096, 4097, 8190, 8194, 12288, 20480, 24576,
             65534, 65536, 65538, 69632, 131072, 131073, 200000]
    mixed = [_weird_chunk(rng, n) for n in edges]
    mixed += [_weird_chunk(rng, int(rng.integers(0, 150000))) for _ in range(24)]
    mix64 = datagen.batch(64, 65536, 0x0E110003)
    small = [_weird_chunk(rng, int(rng.integers(0, 4097))) for _ in range(100)]
    small += datagen.batch(64, 4096, 0x0E110005, mix=datagen.MIX4K)
    big = [datagen.text_chunk(1 << 20, 77) + datagen.binary_chunk(300000, 78)]
    levels = [1, 6, 9, 12] if quick else list(range(13))
    for a in sys.argv[1:]:
        if a.startswith("--levels="):	# a subset, e.g. --levels=6,1 (tools/ab.sh)
            levels = [int(x) for x in a[9:].split(",")]
    bad = 0
    for level in levels:
        fmt = ("deflate", "zlib", "gzip")[level % 3]
        c = api.Compressor(level)
        for name, chunks in (("mixed", mixed), ("mix64", mix64), ("small", small)):
            comps = c.compress_batch_host(fmt, chunks)
            h = hashlib.sha256()
            tot = 0
            for d, z in zip(chunks, comps):
                ok = z is not None and zlib.decompress(z, WB[fmt]) == d
                if not ok:
                    bad += 1
                    print("FAIL", level, name, len(d))
                    continue
                h.update(z)
                tot += len(z)
            print(f"L{level:02d} {fmt:7s} {name:6s} {tot:9d} {h.hexdigest()[:16]}", flush=True)
        z = c.compress(fmt, big[0])
        ok = z is not None and zlib.decompress(z, WB[fmt]) == big[0]
        bad += not ok
        print(f"L{level:02d} {fmt:7s} big    {len(z) if z else 0:9d} "
              f"{hashlib.sha256(z or b'').hexdigest()[:16]}", flush=True)
        c.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
