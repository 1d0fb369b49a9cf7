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
    chunks = datagen.batch(count, size, seed, distinct=distinct,
                           mix=datagen.MIX4K if size <= 4096 else datagen.MIX64K)
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
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--kind", type=int, default=-1, help="use only chunk kind k (0-7) of the mix")
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--fmt", default="gzip")
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
    if a.what in ("inflate", "all"):
        bench_inflate(a, fmt=a.fmt, level=min(a.level, 9))
    if a.what in ("deflate", "all"):
        bench_deflate(a, fmt=a.fmt, level=a.level)


PHASES_DEFLATE = ["init/other", "S0 load", "S2 insert", "S3 round B + worklist", "phase X",
                  "S4 emit", "hist", "S5 codes", "S6 tokens+save", "S6 header", "S5 rank sort", "S5 two trees",
                  "S4 parse", "mc: setup", "mc: merge", "mc: depths",
                  "#RB generations", "mc: clamp+lens", "#RB rounds", "#RB item-generations",
                  "#w0 chain steps", "#w0 evaluate rounds", "S5 precode tree", "S5 precode items",
                  "X: insert (wave N-1)", "X: insert3 (wave N-2)", "X: final parse (wave 0)",
                  "X: round A (wave 1)", "X: wait parse (wave 1)", "X: emit (wave 1)",
                  "X: barrier (wave 1)", "X: wave 0 after parse", "split stats",
                  "X: steps of first parse (wave 1)", "X: wave 0 until first parse done",
                  "RB: generation 0", "RB: generation 1", "RB: generation 2", "RB: generations 3+", "worklist"]


def read_profile(name, labels):
    import ctypes
    from libdeflate_amd import binding
    lib = binding.load()
    try:
        fn = getattr(lib, name)
    except AttributeError:
        return
    buf = (ctypes.c_ulonglong * 40)()
    fn(buf)
    tot = sum(buf)
    if not tot:
        return
    print("  phase cycles (thread 0 of each workgroup, summed):")
    for i, v in enumerate(buf):
        if v:
            lab = labels[i] if i < len(labels) else f"slot{i}"
            if lab.startswith("#"):
                print(f"    {lab:22s} {v:14d}")
            else:
                print(f"    {lab:22s} {v/1e6:12.1f} Mcyc  {100*v/tot:5.1f}%")


def bench_inflate(a, fmt="gzip", level=6):
    from tests import oracle_util, streams
    ref = oracle_util.load_ref()
    distinct = 64
    chunks = datagen.batch(a.chunks, a.size, 0x0E110004, distinct=distinct)
    if a.kind >= 0:
        chunks = [datagen.chunk(a.kind + 8 * (i % 8), a.size, 0x0E110004) for i in range(a.chunks)]
    comp = [ref.compress(fmt, level, c) if ref else streams._zcompress(fmt, level, c)
            for c in chunks[:distinct]]
    if a.kind >= 100:
        # codes of (nearly) one codeword length: bytes drawn evenly from (kind - 100)
        # values, Huffman-only blocks (zlib) - a parse started anywhere never falls
        # in step, the case of par_phase_starts()
        import zlib
        wb = {"deflate": -15, "zlib": 15, "gzip": 31}[fmt]
        # kind >= 1000: the same bytes from (kind - 1000) values through zlib's default strategy
        # (a match now and then among the literals: what a compressor writes over
        # incompressible bytes when it does not store them)
        nv = a.kind - 1000 if a.kind >= 1000 else a.kind - 100
        chunks = [np.random.default_rng(7000 + i).integers(0, nv, a.size, dtype=np.uint8).tobytes()
                  for i in range(distinct)]
        chunks = [chunks[i % distinct] for i in range(a.chunks)]
        comp = []
        for c in chunks[:distinct]:
            co = zlib.compressobj(6, zlib.DEFLATED, wb, 9,
                                  zlib.Z_DEFAULT_STRATEGY if a.kind >= 1000 else zlib.Z_HUFFMAN_ONLY)
            comp.append(co.compress(c) + co.flush())
    offs, blob, sizes = [], bytearray(), []
    for i in range(a.chunks):
        c = comp[i % distinct]
        offs.append(len(blob)); sizes.append(len(c))
        blob += c
        blob += bytes((-len(blob)) % 16)
    blob += bytes(64)
    d_in = torch.frombuffer(blob, dtype=torch.uint8).cuda()
    in_off = torch.tensor(offs, dtype=torch.int64).cuda()
    in_n = torch.tensor(sizes, dtype=torch.int64).cuda()
    d_out = torch.zeros(a.chunks * a.size, dtype=torch.uint8, device="cuda")
    out_off = torch.arange(a.chunks, dtype=torch.int64, device="cuda") * a.size
    out_av = torch.full((a.chunks,), a.size, dtype=torch.int64, device="cuda")
    res = torch.full((a.chunks,), -1, dtype=torch.int32, device="cuda")
    dec = api.Decompressor()
    f = lambda: dec.decompress_batch(fmt, d_in, in_off, in_n, d_out, out_off, out_av, res)
    t = timeit(f, iters=5, warmup=1)
    ok = int((res == 0).sum())
    U = a.chunks * a.size
    C = sum(sizes)
    read_profile("libdeflate_amd_profile_read_inflate",
                 ["hdr", "tables", "tokens", "-", "dec seg(lane0)", "flush seg(lane0)", "rounds(lane0)", "iterations(wave)", "reg-matches", "pipelined", "slow-matches", "slow-bytes", "#par STOP", "#par rounds", "#par EOB rounds", "-", "par: sync", "#sync iterations", "copy: group setup", "copy: slots", "copy: wait for far sources", "copy: flush", "#copy groups", "#slot batches"])
    print(f"inflate[{fmt} L{level}]: {U/t/1e9:.2f} GB/s uncompressed, algorithmic {(U+C)/t/1e9:.2f} GB/s, "
          f"{t*1e3:.2f} ms, ratio {C/U:.3f}, ok {ok}/{a.chunks}")


def bench_deflate(a, fmt="gzip", level=6):
    chunks, data, offs, nb = make_batch(a.chunks, a.size, 0x0E110003)
    if a.kind >= 0:
        one = [datagen.chunk(a.kind + 8 * (i % 8), a.size, 0x0E110003) for i in range(64)]
        chunks = [one[i % 64] for i in range(a.chunks)]
        data = torch.frombuffer(bytearray(b"".join(chunks)), dtype=torch.uint8).cuda()
    n = a.chunks
    c = api.Compressor(level)
    bound = (c.bound(fmt, a.size) + 15) // 16 * 16
    comp = torch.zeros(n * bound, dtype=torch.uint8, device="cuda")
    c_off = torch.arange(n, dtype=torch.int64, device="cuda") * bound
    c_av = torch.full((n,), bound, dtype=torch.int64, device="cuda")
    c_n = torch.zeros(n, dtype=torch.int64, device="cuda")
    f = lambda: c.compress_batch(fmt, data, offs, nb, comp, c_off, c_av, c_n, max_chunk=a.size)
    f(); torch.cuda.synchronize()
    reader = ("libdeflate_amd_profile_read_deflate_small"
              if a.size <= 4096 and level <= 9 and not os.environ.get("LDA_NO_SMALL")
              else "libdeflate_amd_profile_read_deflate")
    read_profile(reader, PHASES_DEFLATE)  # reset
    t = timeit(f, iters=a.iters, warmup=0)
    U = n * a.size
    C = int(c_n.sum())
    print(f"deflate[{fmt} L{level}]: {U/t/1e9:.2f} GB/s uncompressed, algorithmic "
          f"{(U+C)/t/1e9:.2f} GB/s, {t*1e3:.2f} ms, ratio {C/U:.4f}")
    read_profile(reader, PHASES_DEFLATE)


if __name__ == "__main__":
    main()
