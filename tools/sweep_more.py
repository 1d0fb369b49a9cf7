"""Extra seeds of tests/test_deflate_gpu.py::test_random_sweep (one-off confidence run)."""
import sys, os, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libdeflate_amd import api
from tests import test_deflate_gpu as T
from tests import datagen
WB = {"deflate": -15, "zlib": 15, "gzip": 31}
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    for level in (1, 6, 9, 10, 11, 12):
        rng = np.random.default_rng(0xABC000 + 131 * seed + level)
        edges = [4094, 4095, 4096, 4097, 4098, 8190, 8194, 12288, 65534, 65538, 69632, 131071, 135000]
        sizes = [int(rng.choice(edges)) + int(rng.integers(-2, 3)) for _ in range(8)]
        sizes += [int(rng.integers(0, 200000)) for _ in range(16)]
        chunks = [T._weird_chunk(rng, n) for n in sizes]
        fmt = ("deflate", "zlib", "gzip")[(level + seed) % 3]
        c = api.Compressor(level)
        comps = c.compress_batch_host(fmt, chunks)
        d = api.Decompressor()
        for ch, z in zip(chunks, comps):
            ok = z is not None and zlib.decompress(z, WB[fmt]) == ch and len(z) <= c.bound(fmt, len(ch))
            if ok:
                r, _, out = d.decompress(fmt, z, len(ch))
                ok = r == 0 and out == ch
            if not ok:
                bad += 1
                print("FAIL", seed, level, fmt, len(ch))
        c.close(); d.close()
print("done, failures:", bad)
