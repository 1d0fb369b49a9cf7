"""Batched CRC-32 / Adler-32 on the GPU vs the oracle (and zlib as the second
independent control, as programs/test_checksums.c:111-196 does)."""
import zlib

import numpy as np
import pytest

from tests import datagen

pytestmark = pytest.mark.gpu


def _device_batch(chunks, align_pad=0):
    import torch
    offs, sizes, blob = [], [], bytearray()
    for c in chunks:
        blob += bytes(align_pad)
        offs.append(len(blob))
        sizes.append(len(c))
        blob += c
    blob += bytes(64)
    data = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
    return (data, torch.tensor(offs, dtype=torch.int64).cuda(),
            torch.tensor(sizes, dtype=torch.int64).cuda())


@pytest.mark.parametrize("kind", ["crc32", "adler32"])
def test_batch_matches_oracle(kind, oracle):
    import torch
    from libdeflate_amd import api
    rng = np.random.default_rng(7)
    sizes = [0, 1, 2, 15, 16, 17, 63, 64, 65, 1023, 1024, 1025, 4096, 5552,
             5553, 65535, 65536, 65537, 200000]
    chunks = [datagen.chunk(i, n, 0x0E110000) for i, n in enumerate(sizes)]
    chunks.append(b"\xff" * 5553)
    for pad in (0, 1, 7, 13):
        data, offs, nb = _device_batch(chunks, pad)
        inits = rng.integers(0, 2**32, size=len(chunks), dtype=np.uint32)
        if kind == "adler32":
            lo = rng.integers(0, 65521, size=len(chunks))
            hi = rng.integers(0, 65521, size=len(chunks))
            inits = ((hi << 16) | lo).astype(np.uint32)
            inits[-1] = (65520 << 16) | 65520   # test_checksums.c:184-196
        init_t = torch.from_numpy(inits.view(np.int32)).cuda()
        out = torch.zeros(len(chunks), dtype=torch.int32, device="cuda")
        for init in (None, init_t):
            api.checksum_batch(kind, data, offs, nb, out, init=init)
            torch.cuda.synchronize()
            got = out.cpu().numpy().view(np.uint32)
            for i, c in enumerate(chunks):
                iv = int(inits[i]) if init is not None else (0 if kind == "crc32" else 1)
                want = getattr(oracle, kind)(c, iv)
                assert got[i] == want, (kind, pad, i, len(c))
                z = zlib.crc32(c, iv) if kind == "crc32" else zlib.adler32(c, iv)
                assert want == z


def test_single_buffer_api(oracle):
    from libdeflate_amd import api, binding
    lib = binding.load()
    # NULL-buffer rules, lib/crc32.c:259-260, lib/adler32.c:159-160
    assert lib.libdeflate_crc32(1234, None, 1234) == 0
    assert lib.libdeflate_adler32(1234, None, 0) == 1
    data = datagen.text_chunk(100000, 3)
    assert api.crc32(data) == oracle.crc32(data)
    assert api.adler32(data) == oracle.adler32(data)
    # chaining: f(f(v,A),B) == f(v,A||B)  (test_checksums.c:73-84)
    a, b = data[:33333], data[33333:]
    assert api.crc32(b, api.crc32(a)) == api.crc32(data)
    assert api.adler32(b, api.adler32(a)) == api.adler32(data)
    assert api.crc32(b"") == 0 and api.adler32(b"") == 1
