#!/usr/bin/env python3
"""bench_single_api.py [MiB per call] [calls] - the reference's single-buffer
calls (libdeflate_gzip_compress / libdeflate_gzip_decompress, host pointers in
and out) on enwik-style text, one call at a time the way programs/benchmark.c
drives them (default 1 MiB chunks; like it, the buffers are allocated once and
reused): MB/s host to host, for DESIGN.md's note on large single buffers.  A
tuning aid, not part of the product."""
import os
import sys
import time
from ctypes import byref, c_size_t, c_void_p

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from tests import datagen  # noqa: E402


def main():
    from libdeflate_amd import api, binding
    mib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    level = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    n = int(mib * (1 << 20))
    ndist = min(calls, 4)
    chunks = [np.frombuffer(datagen.text_chunk(n, 0x0E110100 + i), dtype=np.uint8)
              for i in range(ndist)]
    c, d = api.Compressor(level), api.Decompressor()
    lib = binding.load()
    bound = c.bound("gzip", n)
    zbuf = [np.zeros(bound, dtype=np.uint8) for _ in range(ndist)]
    back = np.zeros(n, dtype=np.uint8)
    P = lambda a: a.ctypes.data_as(c_void_p)
    zn = [0] * ndist
    for i in range(ndist):      # warm-up: device buffers, pinned staging, page faults
        zn[i] = lib.libdeflate_gzip_compress(c._h, P(chunks[i]), n, P(zbuf[i]), bound)
        assert zn[i]
    t0 = time.perf_counter()
    for k in range(calls):
        i = k % ndist
        zn[i] = lib.libdeflate_gzip_compress(c._h, P(chunks[i]), n, P(zbuf[i]), bound)
    t1 = time.perf_counter()
    ao = c_size_t(0)
    for i in range(ndist):
        assert lib.libdeflate_gzip_decompress(d._h, P(zbuf[i]), zn[i], P(back), n, byref(ao)) == 0
        assert ao.value == n and np.array_equal(back, chunks[i])
    t2 = time.perf_counter()
    for k in range(calls):
        i = k % ndist
        lib.libdeflate_gzip_decompress(d._h, P(zbuf[i]), zn[i], P(back), n, byref(ao))
    t3 = time.perf_counter()
    U = n * calls
    print(f"{mib:g} MiB per call x {calls} (level {level}): compress {U / (t1 - t0) / 1e6:.0f} MB/s "
          f"({(t1 - t0) / calls * 1e3:.2f} ms per call), decompress {U / (t3 - t2) / 1e6:.0f} MB/s "
          f"({(t3 - t2) / calls * 1e3:.2f} ms per call), ratio {sum(zn) / (n * ndist):.4f}, "
          f"stream path: {binding.stream_stats()['parallel']}")


if __name__ == "__main__":
    main()
