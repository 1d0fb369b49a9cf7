#!/usr/bin/env python3
"""bench_stream.py [MiB ...] - libdeflate_gzip_decompress on ONE reference-
compressed stream per call (host pointers in and out), the shape of
programs/gzip.c:187-303: GB/s host to host and the host-side phase times of the
many-wave path (libdeflate_amd_stream_stats).  A tuning aid."""
import os
import sys
import time
from ctypes import byref, c_size_t, c_void_p

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from tests import datagen, oracle_util, streams  # noqa: E402


def main():
    from libdeflate_amd import api, binding
    sizes = [float(a) for a in sys.argv[1:] if not a.startswith("-")] or [1, 4, 16, 64]
    kind = "mix" if "--mix" in sys.argv else "text"
    # --kind=K: chunks of kind K of the mix only (5 binary counters, 6 16-symbol text, 7 random)
    for a in sys.argv:
        if a.startswith("--kind="):
            kind = "kind" + a.split("=")[1]
    # --stored: level 0 (stored blocks only); --fixed: zlib Z_FIXED (static blocks only)
    # --huff: zlib Z_HUFFMAN_ONLY over bytes drawn evenly from 250 values (dynamic blocks of
    # literals under a code of nearly one codeword length: parses never fall in step)
    shape = ("stored" if "--stored" in sys.argv else "fixed" if "--fixed" in sys.argv else
             "huff" if "--huff" in sys.argv else "dynamic")
    level = 6
    ref = oracle_util.load_ref()
    d = api.Decompressor()
    lib = binding.load()
    for mib in sizes:
        n = int(mib * (1 << 20))
        if kind == "text":
            data = datagen.text_chunk(n, 0x0E110200)
        elif kind.startswith("kind"):
            k = int(kind[4:])
            data = b"".join(datagen.chunk(k + 8 * i, 65536, 0x0E110200) for i in range((n + 65535) >> 16))[:n]
        else:
            data = b"".join(datagen.chunk(i, 65536, 0x0E110200) for i in range((n + 65535) >> 16))[:n]
        if shape == "huff":
            import zlib
            nv = int(os.environ.get("HUFF_VALUES", "250"))
            data = np.random.default_rng(0x0E11).integers(0, nv, n, dtype=np.uint8).tobytes()
            co = zlib.compressobj(6, zlib.DEFLATED, 31, 9, zlib.Z_HUFFMAN_ONLY)
            z = co.compress(data) + co.flush()
        elif shape == "stored":
            z = streams._zcompress("gzip", 0, data)
        elif shape == "fixed":
            import zlib
            co = zlib.compressobj(6, zlib.DEFLATED, 31, 9, zlib.Z_FIXED)
            z = co.compress(data) + co.flush()
        else:
            z = ref.compress("gzip", level, data) if ref else streams._zcompress("gzip", level, data)
        zin = np.frombuffer(z, dtype=np.uint8)
        out = np.zeros(n, dtype=np.uint8)
        best, st = 1e9, None
        for _ in range(6):
            ao = c_size_t(0)
            t0 = time.perf_counter()
            r = lib.libdeflate_gzip_decompress(d._h, zin.ctypes.data_as(c_void_p), zin.size,
                                               out.ctypes.data_as(c_void_p), n, byref(ao))
            dt = time.perf_counter() - t0
            assert r == 0 and ao.value == n
            if dt < best:
                best, st = dt, binding.stream_stats()
        assert out.tobytes() == data
        ph = {k: v for k, v in st.items() if k.startswith("us_")}
        print(f"{mib:g} MiB {kind} {shape} L{level}: {best * 1e3:.2f} ms = {n / best / 1e9:.2f} GB/s "
              f"parallel={st['parallel']} chunks={st['chunks_decoded']} repairs={st['repairs']} {ph}")


if __name__ == "__main__":
    main()
