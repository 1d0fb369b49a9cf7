#!/usr/bin/env python3
"""bench_single_api.py [MiB per call] [calls] - the reference's single-buffer
calls (libdeflate_gzip_compress / libdeflate_gzip_decompress, host pointers in
and out) on enwik-style text, one call at a time the way programs/benchmark.c
drives them (default 1 MiB chunks): MB/s host to host, for DESIGN.md's note on
large single buffers.  A tuning aid, not part of the product."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from tests import datagen  # noqa: E402


def main():
    from libdeflate_amd import api
    mib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    n = int(mib * (1 << 20))
    chunks = [datagen.text_chunk(n, 0x0E110100 + i) for i in range(calls)]
    c, d = api.Compressor(6), api.Decompressor()
    comp = [c.compress("gzip", x) for x in chunks[:2]]      # warm-up
    t0 = time.perf_counter()
    comp = [c.compress("gzip", x) for x in chunks]
    t1 = time.perf_counter()
    back = [d.decompress("gzip", z, n) for z in comp[:1]]
    t2 = time.perf_counter()
    back = [d.decompress("gzip", z, n) for z in comp]
    t3 = time.perf_counter()
    assert all(b[0] == 0 and b[-1] == x for b, x in zip(back, chunks))
    U = n * calls
    print(f"{mib} MiB per call x {calls}: compress {U / (t1 - t0) / 1e6:.0f} MB/s, "
          f"decompress {U / (t3 - t2) / 1e6:.1f} MB/s, ratio {sum(map(len, comp)) / U:.4f}")


if __name__ == "__main__":
    main()
