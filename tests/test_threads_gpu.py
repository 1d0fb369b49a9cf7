"""The threading contract of libdeflate.h:56-57,178-179: one object must not be
used by two threads at once, DIFFERENT objects may be - the reference's own
CPU harness (one compressor + decompressor per thread) relies on it.  Four
host threads, each with its own compressor and decompressor, mix single-buffer
calls, the segmented large-buffer path and the host-pointer batches for a few
seconds; every result is checked with the real reference (zlib where
oracle/_ref did not travel).  ctypes releases the GIL around the calls, so the
library really is entered concurrently."""
import threading
import time
import zlib

import numpy as np
import pytest

from tests import datagen, oracle_util

pytestmark = pytest.mark.gpu
WBITS = {"deflate": -15, "zlib": 15, "gzip": 31}


def _worker(tid, seconds, errors, counts):
    try:
        from libdeflate_amd import api
        ref = oracle_util.load_ref()     # per thread: the Ref object is not shared
        rng = np.random.default_rng(1000 + tid)
        level = (1, 6, 9, 12)[tid % 4]
        c, d = api.Compressor(level), api.Decompressor()

        def ref_inflate(fmt, z, n):
            if ref is not None:
                r = ref.decompress_ex(fmt, z, n)
                assert r[0] == 0 and r[1] == len(z), (tid, "reference rejects", r[:3])
                return r[3]
            return zlib.decompress(z, WBITS[fmt])

        def ref_deflate(fmt, data):
            if ref is not None:
                return ref.compress(fmt, 6, data)
            co = zlib.compressobj(6, zlib.DEFLATED, WBITS[fmt])
            return co.compress(data) + co.flush()

        t_end = time.time() + seconds
        it = 0
        while time.time() < t_end:
            fmt = ("deflate", "zlib", "gzip")[it % 3]
            kind = it % 4
            seed = 0x7EAD0000 + 4096 * tid + it
            if kind == 0:       # single buffer, below the segment threshold
                data = datagen.chunk(it, int(rng.integers(0, 100000)), seed)
                z = c.compress(fmt, data)
                assert z is not None and ref_inflate(fmt, z, len(data)) == data
                assert d.decompress_ex(fmt, z, len(data)) == (0, len(z), len(data), data)
            elif kind == 1:     # one large buffer: compress_large
                data = datagen.chunk(it, int(rng.integers(200000, 900000)), seed)
                z = c.compress(fmt, data)
                assert z is not None and ref_inflate(fmt, z, len(data)) == data
                assert d.decompress_ex(fmt, z, len(data)) == (0, len(z), len(data), data)
            elif kind == 2:     # host-pointer batch, compress side
                chunks = [datagen.chunk(i, int(rng.integers(1, 70000)), seed + i)
                          for i in range(12)]
                zs = c.compress_batch_host(fmt, chunks)
                for data, z in zip(chunks, zs):
                    assert z is not None and ref_inflate(fmt, z, len(data)) == data
            else:               # host-pointer batch, decompress side, reference streams
                chunks = [datagen.chunk(i, int(rng.integers(1, 70000)), seed + i)
                          for i in range(12)]
                zs = [ref_deflate(fmt, x) for x in chunks]
                got = d.decompress_batch_host(fmt, zs, [len(x) for x in chunks])
                for data, z, g in zip(chunks, zs, got):
                    assert g == (0, len(z), len(data), data)
                assert api.crc32(chunks[0]) == zlib.crc32(chunks[0])
                assert api.adler32(chunks[1]) == zlib.adler32(chunks[1])
            it += 1
        counts[tid] = it
        c.close()
        d.close()
    except BaseException as e:      # noqa: BLE001 - reported by the main thread
        import traceback
        errors.append((tid, repr(e), traceback.format_exc()[-1500:]))


def test_four_threads_four_objects():
    nthreads, seconds = 4, 4.0
    errors, counts = [], [0] * nthreads
    ts = [threading.Thread(target=_worker, args=(t, seconds, errors, counts))
          for t in range(nthreads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ts), "a worker hangs"
    assert not errors, errors
    assert all(n >= 4 for n in counts), counts      # every kind of call at least once
    print("iterations per thread:", counts)
