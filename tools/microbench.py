"""Per-kernel timing on one GPU (HIP events via torch on the launch stream).
Usage: python tools/microbench.py [crc32|adler32|inflate|deflate|all] [--chunks N] [--size S]
Prints achieved algorithmic GB/s per kernel; used while tuning, not by the driver."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from libdeflate_amd import api
from tests import datagen


def make_batch(count, size, seed, distinct=64):
    chunks = datagen.batch(count, size, seed, distinct=distinct)
    blob = b"".join(chunks)
    data = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
    offs = torch.arange(count, dtype=torch.int64, device="cuda") * size
    nb = torch.full((count,), size, dtype=torch.int64, device="cuda")
    return chunks, data, offs, nb


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters / 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--chunks", type=int, default=16384)
    ap.add_argument("--size", type=int, default=65536)
    a = ap.parse_args()
    chunks, data, offs, nb = make_batch(a.chunks, a.size, 0x0E110003)
    total = a.chunks * a.size
    if a.what in ("crc32", "adler32", "all"):
        out = torch.zeros(a.chunks, dtype=torch.int32, device="cuda")
        for kind in ("crc32", "adler32"):
            if a.what not in (kind, "all"):
                continue
            t = timeit(lambda: api.checksum_batch(kind, data, offs, nb, out))
            print(f"{kind}: {total/t/1e9:.1f} GB/s  ({t*1e3:.3f} ms for {total/2**20:.0f} MiB)")


if __name__ == "__main__":
    main()
