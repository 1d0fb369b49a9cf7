import sys, time
import numpy as np
from ctypes import c_void_p
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from tests import datagen
from libdeflate_amd import api, binding
mib = float(sys.argv[1]); n = int(mib * (1 << 20))
chunks = [np.frombuffer(datagen.text_chunk(n, 0x0E110100 + i), dtype=np.uint8) for i in range(2)]
c = api.Compressor(6); lib = binding.load(); bound = c.bound("gzip", n)
zbuf = [np.zeros(bound, dtype=np.uint8) for _ in range(2)]
P = lambda a: a.ctypes.data_as(c_void_p)
for i in range(2):
    lib.libdeflate_gzip_compress(c._h, P(chunks[i]), n, P(zbuf[i]), bound)
ts = []
for k in range(10):
    t0 = time.perf_counter(); lib.libdeflate_gzip_compress(c._h, P(chunks[k % 2]), n, P(zbuf[k % 2]), bound); ts.append(time.perf_counter() - t0)
print(f"{mib:g} MiB compress: best {min(ts)*1e3:.2f} ms, median {sorted(ts)[5]*1e3:.2f} ms, all " + " ".join(f"{t*1e3:.1f}" for t in ts))
