#!/usr/bin/env python3
"""try_level.py LEVEL SIZE [COUNT] - compress COUNT chunks of SIZE bytes of the
benchmark mix at LEVEL through the host-pointer batch call and decode them
with zlib: the smallest possible "does this build still work" run, meant to be
started under `timeout` when a change to the compress kernel's wave hand-overs
could hang it."""
import sys, zlib
sys.path.insert(0, '/root/repo')
from libdeflate_amd import api
from tests import datagen
L = int(sys.argv[1]); n = int(sys.argv[2]); cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 1
c = api.Compressor(L)
chunks = [datagen.chunk(i, n, 0x0E110003) for i in range(cnt)]
z = c.compress_batch_host("deflate", chunks)
ok = all(zlib.decompress(a, -15) == b for a, b in zip(z, chunks))
print("L", L, "n", n, "cnt", cnt, "ok", ok, sum(map(len, z)))
