"""Decode N reference-compressed chunks on the GPU and compare every byte.
Usage: python tools/check_inflate.py [N] [size]   (LDA_INFLATE_PAR=0 selects lane-per-stream)"""
import os, sys
sys.path.insert(0, '/root/repo')
import torch
from libdeflate_amd import api
from tests import datagen, oracle_util
ref = oracle_util.load_ref()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
size = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
chunks = [datagen.chunk(i, size, 0x0E110004) for i in range(n)]
comp = [ref.compress('gzip', 6, c) for c in chunks]
blob = bytearray(); offs=[]; sizes=[]
for c in comp:
    offs.append(len(blob)); sizes.append(len(c)); blob += c; blob += bytes((-len(blob)) % 16)
blob += bytes(256)
d_in = torch.frombuffer(blob, dtype=torch.uint8).cuda()
in_off = torch.tensor(offs, dtype=torch.int64).cuda(); in_n = torch.tensor(sizes, dtype=torch.int64).cuda()
out = torch.zeros(n*size + 4096, dtype=torch.uint8, device='cuda')
out_off = torch.arange(n, dtype=torch.int64, device='cuda')*size
out_av = torch.full((n,), size, dtype=torch.int64, device='cuda')
res = torch.full((n,), -1, dtype=torch.int32, device='cuda')
dec = api.Decompressor()
dec.decompress_batch('gzip', d_in, in_off, in_n, out, out_off, out_av, res)
torch.cuda.synchronize()
r = res.cpu().tolist()
o = out.cpu().numpy().tobytes()
for i in range(n):
    exp = chunks[i]; got = o[i*size:(i+1)*size]
    if r[i] != 0 or got != exp:
        first = next((k for k in range(size) if got[k] != exp[k]), -1)
        print('stream', i, 'result', r[i], 'first mismatch at', first)
        if first >= 0:
            print('  exp', exp[first-8:first+24]); print('  got', got[first-8:first+24])
print('done', sum(1 for x in r if x == 0), '/', n)
