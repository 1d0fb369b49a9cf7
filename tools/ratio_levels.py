"""Compressed sizes per level and content kind, ours vs the reference build in oracle/_ref
(run on the GPU box: python tools/ratio_levels.py [levels...])."""
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libdeflate_amd import api
from tests import datagen, oracle_util

levels = [int(a) for a in sys.argv[1:]] or [9, 10, 11, 12]
ref = oracle_util.load_ref()
for size, seed, mix in ((65536, 0x0E110003, datagen.MIX64K), (4096, 0x0E110005, datagen.MIX4K)):
    chunks = datagen.batch(16, size, seed, distinct=16, mix=mix)
    for lvl in levels:
        c = api.Compressor(lvl)
        comps = c.compress_batch_host("deflate", chunks)
        c.close()
        for d, z in zip(chunks, comps):
            assert zlib.decompress(z, -15) == d, (lvl, size)
        kinds = {}
        for i, (d, z) in enumerate(zip(chunks, comps)):
            k = mix[i % 8].__name__
            o, r = kinds.get(k, (0, 0))
            kinds[k] = (o + len(z), r + (len(ref.compress("deflate", lvl, d)) if ref else 0))
        tot_o = sum(v[0] for v in kinds.values())
        tot_r = sum(v[1] for v in kinds.values())
        detail = "  ".join(f"{k[:-6]} {o}/{r}" for k, (o, r) in kinds.items())
        print(f"size {size} level {lvl}: ours {tot_o} ref {tot_r} ({tot_o / max(tot_r, 1):.4f})  {detail}")
