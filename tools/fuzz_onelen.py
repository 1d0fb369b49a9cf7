#!/usr/bin/env python3
"""fuzz_onelen.py [seeds] - streams whose blocks have literal codewords of (nearly) one length
through libdeflate_*_decompress: bytes drawn from N values (N random), zlib's Huffman-only,
default and filtered strategies, repeated stretches mixed in (matches across the planned
starts), text between them.  Every stream must come back byte-exact THROUGH the many-wave
path: a disagreement between phase_count() and chunk_run() (csrc/inflate_stream.hip) does not
corrupt anything - the decode pass notices and the sequential kernel answers - but it shows
here as parallel == 0.  A tuning / checking aid."""
import sys
import zlib

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from libdeflate_amd import api, binding  # noqa: E402
from tests import datagen  # noqa: E402


def main():
    nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    d = api.Decompressor()
    bad = seq = 0
    for seed in range(nseeds):
        rng = np.random.default_rng(0x1E60 + seed)
        parts = []
        total = 0
        while total < (1 << 20) + int(rng.integers(0, 3 << 20)):
            kind = int(rng.integers(0, 4))
            n = int(rng.integers(20000, 200000))
            if kind == 0:
                b = datagen.text_chunk(n, 100 + seed)
            else:
                nv = int(rng.choice([16, 32, 64, 128, 200, 250, 255, 256]))
                b = rng.integers(0, nv, n, dtype=np.uint8).tobytes()
                if kind == 2:   # repeats: matches among the literals
                    k = int(rng.integers(1, 40))
                    for _ in range(k):
                        a = int(rng.integers(0, n - 400))
                        ln = int(rng.integers(3, 300))
                        at = int(rng.integers(a + 1, min(n - ln, a + 30000)))
                        b = b[:at] + b[a:a + ln] + b[at + ln:]
            parts.append(b)
            total += len(b)
        data = b"".join(parts)
        strat = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_HUFFMAN_ONLY, zlib.Z_FILTERED][seed % 3]
        co = zlib.compressobj(int(rng.integers(1, 10)), zlib.DEFLATED, -15, 9, strat)
        z = co.compress(data) + co.flush()
        r = d.decompress_ex("deflate", z, len(data))
        st = binding.stream_stats()
        ok = r[0] == 0 and r[3] == data
        bad += not ok
        seq += st["parallel"] == 0
        if not ok or st["parallel"] == 0:
            print(f"seed {seed}: ok={ok} {st}")
    print(f"{nseeds} streams, {bad} wrong, {seq} answered by the sequential kernel")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
