import sys; sys.path.insert(0,'/root/repo')
from tests import datagen, oracle_util
from tools import stream_stats as ss
from libdeflate_amd import api
ref = oracle_util.load_ref()
c = api.Compressor(6)
for idx in (0, 5, 6):
    d = datagen.chunk(idx, 65536, 0x0E110003)
    ours = c.compress("deflate", d); theirs = ref.compress("deflate", 6, d)
    print("chunk", idx, datagen.MIX64K[idx % 8].__name__)
    assert ss.summarize(" ours", ours) == d
    assert ss.summarize(" ref ", theirs) == d
