#!/usr/bin/env python3
"""fuzz_deflate.py [seeds...] - a longer-running companion of
tests/test_deflate_gpu.py::test_random_sweep: seeded sizes around the tile /
block / segment limits and contents that stress the parsers, every level and
format, through the GPU compressor (64 KiB kernel, small-buffer kernel and the
single-buffer API's segmented path); every stream is decoded with zlib and
checked against compress_bound.  Exits non-zero on the first failure."""
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from tests.test_deflate_gpu import _weird_chunk  # noqa: E402

WB = {"deflate": -15, "zlib": 15, "gzip": 31}


def run(seeds, budget_s=None, log=print):
    """-> (buffers compressed, failures); stops after `budget_s` seconds (at
    least one seed)."""
    import time
    from libdeflate_amd import api
    edges = [1, 2, 3, 17, 18, 19, 52, 53, 4094, 4095, 4096, 4097, 8190, 8194, 12288,
             65534, 65536, 65538, 69632, 131071, 131072, 131073]
    nbad = nbuf = 0
    t0 = time.time()
    for k, seed in enumerate(seeds):
        if budget_s is not None and k and time.time() - t0 > budget_s:
            break
        rng = np.random.default_rng(0x0F220000 + seed)
        level = int(rng.integers(0, 13))
        fmt = ("deflate", "zlib", "gzip")[seed % 3]
        small = seed % 4 == 0           # a batch the small-buffer kernel takes
        if small:
            sizes = [int(rng.integers(0, 4097)) for _ in range(200)]
            level = min(level, 9)
        else:
            sizes = [max(0, int(rng.choice(edges)) + int(rng.integers(-2, 3))) for _ in range(12)]
            sizes += [int(rng.integers(0, 200000)) for _ in range(12)]
        chunks = [_weird_chunk(rng, n) for n in sizes]
        c = api.Compressor(level)
        comps = c.compress_batch_host(fmt, chunks)
        if not small:                   # and the single-buffer API on a few
            for d in chunks[:3]:
                chunks.append(d)
                comps.append(c.compress(fmt, d))
        for d, z in zip(chunks, comps):
            ok = z is not None and len(z) <= c.bound(fmt, len(d)) and zlib.decompress(z, WB[fmt]) == d
            if not ok:
                nbad += 1
                log("FAIL", seed, level, fmt, len(d), None if z is None else len(z))
        c.close()
        nbuf += len(chunks)
        log(f"seed {seed}: level {level} {fmt} {'small' if small else 'mixed'} "
            f"{len(chunks)} buffers, {sum(map(len, chunks))} bytes, bad {nbad}", flush=True)
        if nbad:
            break
    return nbuf, nbad


def main():
    seeds = [int(a) for a in sys.argv[1:]] or [1, 2, 3]
    _, bad = run(seeds)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
