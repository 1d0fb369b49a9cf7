import os, sys
sys.path.insert(0, '/root/repo')
import torch
from libdeflate_amd import api
from tests import datagen
# warm-up: a compress call (its kernel uses scratch memory)
c = api.Compressor(6)
data = datagen.chunk(0, 65536, 1)
out = c.compress('gzip', data)
print('warm-up compress', len(out))
exec(open(os.path.join(os.path.dirname(__file__), 'dbg_inflate.py')).read())
