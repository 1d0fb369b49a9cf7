#!/usr/bin/env python3
"""bench_host_batch.py - the host-pointer batch entry points
(libdeflate_amd_{compress,decompress}_batch_host: what a cgo / JNI caller
holding ordinary buffers uses) on 4096 x 64 KiB of the benchmark mix, from
pageable host memory and back, warm (the first call allocates staging):
MB/s per direction.  A tuning aid; LIBDEFLATE_AMD_LIB selects the build."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from tests import datagen  # noqa: E402


def main():
    from libdeflate_amd import api, binding
    lib = binding.load()
    n, size = 4096, 65536
    level = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    blob = np.frombuffer(b"".join(datagen.batch(n, size, 0x0E110003, distinct=256)), dtype=np.uint8).copy()
    c, d = api.Compressor(level), api.Decompressor()
    bound = c.bound("gzip", size)
    comp = np.empty(n * bound, dtype=np.uint8)
    back = np.empty(n * size, dtype=np.uint8)
    P, SZ = ctypes.c_void_p, ctypes.c_size_t
    inp = (P * n)(*[blob.ctypes.data + i * size for i in range(n)])
    inn = (SZ * n)(*([size] * n))
    outp = (P * n)(*[comp.ctypes.data + i * bound for i in range(n)])
    outa = (SZ * n)(*([bound] * n))
    outn = (SZ * n)()
    res = (ctypes.c_int32 * n)()
    ain = (SZ * n)()
    backp = (P * n)(*[back.ctypes.data + i * size for i in range(n)])
    tc = td = 1e9
    for it in range(4):
        t0 = time.perf_counter()
        rc = lib.libdeflate_amd_compress_batch_host(c._h, 2, n, inp, inn, outp, outa, outn)
        t1 = time.perf_counter()
        assert rc == 0
        rc = lib.libdeflate_amd_decompress_batch_host(d._h, 2, n, outp, outn, backp, inn, res, ain, None)
        t2 = time.perf_counter()
        assert rc == 0 and not any(res)
        if it:
            tc, td = min(tc, t1 - t0), min(td, t2 - t1)
    assert np.array_equal(back, blob)
    U = n * size
    print(f"host-pointer batch, level {level}: compress {U / tc / 1e6:.0f} MB/s, "
          f"decompress {U / td / 1e6:.0f} MB/s, ratio {sum(outn) / U:.4f}")


if __name__ == "__main__":
    main()
