"""GPU compressor (HIP) through the C-ABI.  Compressed bytes are unpinned by
the reference (libdeflate.h:76-83); what is checked is what the reference
guarantees and its own tests check (scripts/exec_tests.sh:23-36,
test_litrunlen_overflow.c, test_trailing_bytes.c):
  - the stream is valid and round-trips byte-exact through the ORACLE decoder
    and through zlib (independent control),
  - size <= compress_bound, container bytes/footers exact,
  - returns 0 exactly when the output does not fit,
  - ratio tracks the reference's at the same level (reported, loose gate)."""
import os
import zlib

import numpy as np
import pytest

from tests import datagen, streams

pytestmark = pytest.mark.gpu
WBITS = {"deflate": -15, "zlib": 15, "gzip": 31}


_REF = []


def _check_roundtrip(oracle, fmt, data, comp, tag):
    """GPU stream -> the restated oracle, the REAL reference (oracle/_ref)
    and zlib all return the original bytes and consume the whole stream."""
    assert comp is not None, tag
    if not _REF:
        from tests import oracle_util
        _REF.append(oracle_util.load_ref())
    for name, chk in (("oracle", oracle), ("reference", _REF[0])):
        if chk is None:
            continue
        r, ain, aout, out = chk.decompress_ex(fmt, comp, len(data))
        assert r == 0, (tag, name, "result", r)
        assert ain == len(comp), (tag, name, "trailing bytes in stream")
        assert out == data, (tag, name, "round trip differs")
    assert zlib.decompress(comp, WBITS[fmt]) == data, (tag, "zlib control")


@pytest.mark.parametrize("level", [0, 1, 2, 4, 5, 6, 8, 9, 10, 11, 12])
def test_roundtrip_sizes_formats(level, oracle):
    from libdeflate_amd import api
    c = api.Compressor(level)
    sizes = [0, 1, 18, 19, 20, 31, 32, 51, 52, 511, 512, 4999, 5000, 5001,
             65535, 65536, 65537, 300000]
    for fmt in ("deflate", "zlib", "gzip"):
        chunks = [datagen.chunk(i, n, 0x0E110020 + level) for i, n in enumerate(sizes)]
        comps = c.compress_batch_host(fmt, chunks)
        for d, z in zip(chunks, comps):
            _check_roundtrip(oracle, fmt, d, z, (level, fmt, len(d)))
            assert len(z) <= c.bound(fmt, len(d))
    c.close()


def _weird_chunk(rng, n):
    """Inputs built to stress the parsers: periodic data with long matches,
    matches that straddle tile (4096) and block ends, switches of content."""
    kind = int(rng.integers(0, 7))
    if kind == 0:
        return bytes(n)
    if kind == 1:       # short period: every match 258 long at tiny distances
        per = bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
        return (per * (n // len(per) + 1))[:n]
    if kind == 2:       # long period just inside / outside the window
        per = bytes(rng.integers(0, 256, int(rng.choice([4090, 4096, 4100, 28000, 33000])), dtype=np.uint8))
        return (per * (n // len(per) + 1))[:n]
    if kind == 3:       # text with random bytes sprinkled in
        d = bytearray(datagen.text_chunk(n, int(rng.integers(1 << 30))))
        for i in rng.integers(0, max(n, 1), n // 50):
            d[i] = int(rng.integers(0, 256))
        return bytes(d)
    if kind == 4:       # pieces of different content, cut at odd places
        out = bytearray()
        while len(out) < n:
            k = int(rng.integers(0, 8))
            out += datagen.chunk(k, int(rng.integers(1, 20000)), int(rng.integers(1 << 30)))
        return bytes(out[:n])
    if kind == 5:       # few symbols
        return bytes(rng.integers(0, 3, n, dtype=np.uint8) + 65)
    return datagen.chunk(int(rng.integers(0, 8)), n, int(rng.integers(1 << 30)))


@pytest.mark.parametrize("level", list(range(13)))
def test_random_sweep(level, oracle):
    """Seeded sweep over sizes around the tile / block / segment limits and
    contents that stress every parser (levels 10-12: the min-cost parse and
    its re-parse of a block's first tile)."""
    from libdeflate_amd import api
    rng = np.random.default_rng(0x0E110060 + level)
    edges = [4094, 4095, 4096, 4097, 4098, 8190, 8194, 12288, 65534, 65538, 69632, 131071]
    sizes = [int(rng.choice(edges)) + int(rng.integers(-2, 3)) for _ in range(10)]
    sizes += [int(rng.integers(0, 150000)) for _ in range(14)]
    chunks = [_weird_chunk(rng, n) for n in sizes]
    fmt = ("deflate", "zlib", "gzip")[level % 3]
    c = api.Compressor(level)
    comps = c.compress_batch_host(fmt, chunks)
    for d, z in zip(chunks, comps):
        _check_roundtrip(oracle, fmt, d, z, (level, fmt, len(d)))
        assert len(z) <= c.bound(fmt, len(d))
    c.close()


def test_single_buffer_api_and_overflow(oracle):
    from libdeflate_amd import api
    c = api.Compressor(6)
    data = streams.trailing_bytes_input()
    for fmt in ("deflate", "zlib", "gzip"):
        z = c.compress(fmt, data)
        _check_roundtrip(oracle, fmt, data, z, fmt)
        # exact fit succeeds, one byte less returns 0 (libdeflate.h:73-74)
        assert c.compress(fmt, data, len(z)) is not None
        assert c.compress(fmt, data, len(z) - 1) is None
    rnd = datagen.random_chunk(70000, 5)
    z = c.compress("deflate", rnd)
    assert len(z) <= c.bound("deflate", len(rnd))
    _check_roundtrip(oracle, "deflate", rnd, z, "random")
    assert c.compress("gzip", b"abc", 18) is None     # gzip_compress.c:41-42
    assert c.compress("zlib", b"abc", 6) is None      # zlib_compress.c:42-43
    lit = streams.litrunlen_input()                   # test_litrunlen_overflow.c
    for lvl in (3, 6, 12):
        cl = api.Compressor(lvl)
        _check_roundtrip(oracle, "deflate", lit, cl.compress("deflate", lit), lvl)
        cl.close()
    c.close()


def test_container_bytes(oracle):
    """header/footer bytes per lib/gzip_compress.c:44-79, zlib_compress.c:45-72"""
    from libdeflate_amd import api
    data = datagen.text_chunk(10000, 9)
    want_zlib = {1: b"\x78\x01", 5: b"\x78\x5e", 6: b"\x78\x9c", 9: b"\x78\xda"}
    want_xfl = {1: 4, 6: 0, 9: 2}
    for lvl in (1, 5, 6, 9):
        c = api.Compressor(lvl)
        z = c.compress("zlib", data)
        assert z[:2] == want_zlib[lvl]
        assert int.from_bytes(z[-4:], "big") == zlib.adler32(data)
        if lvl in want_xfl:
            g = c.compress("gzip", data)
            assert g[:10] == b"\x1f\x8b\x08\x00\x00\x00\x00\x00" + bytes([want_xfl[lvl], 0xFF])
            assert int.from_bytes(g[-8:-4], "little") == zlib.crc32(data)
            assert int.from_bytes(g[-4:], "little") == len(data)
        c.close()


def test_ratio_tracks_reference(oracle):
    """GPU level L against the reference's compressed size at level L on the
    64 KiB mix, gated where the documents claim it: within
    1.2 % at levels 1 / 6 / 9 and 1.0 % at levels 10 / 12 (measured: 1.004 -
    1.007x; the gate was 3 % until round 5, which a 2 % regression would have
    passed)."""
    from libdeflate_amd import api
    from tests import oracle_util
    ref = oracle_util.load_ref()
    chunks = [datagen.chunk(i, 65536, 0x0E110003) for i in range(16)]
    sizes = {}
    for lvl in (1, 6, 9, 10, 12):
        c = api.Compressor(lvl)
        comps = c.compress_batch_host("deflate", chunks)
        ours = sizes[lvl] = sum(len(z) for z in comps)
        for d, z in zip(chunks, comps):
            _check_roundtrip(oracle, "deflate", d, z, lvl)
        if ref:
            theirs = sum(len(ref.compress("deflate", lvl, d)) for d in chunks)
        else:
            theirs = sum(len(streams._zcompress("deflate", lvl, d)) for d in chunks)
        print(f"level {lvl}: ours {ours} ref {theirs} ratio {ours/theirs:.4f}")
        assert ours <= theirs * (1.012 if lvl <= 9 else 1.010), (lvl, ours, theirs)
        c.close()
    # the min-cost parse of levels 10-12 must pay for itself
    assert sizes[10] < sizes[9] and sizes[12] <= sizes[10]


def test_device_batch_roundtrip(oracle):
    """Config 2/3 shape at reduced count: compress in HBM, decompress in HBM
    with our own decoder, compare every byte; sizes <= bound."""
    import torch
    from libdeflate_amd import api
    n, size = 256, 65536
    chunks = datagen.batch(n, size, 0x0E110002, distinct=32)
    data = torch.frombuffer(bytearray(b"".join(chunks)), dtype=torch.uint8).cuda()
    for fmt, lvl in (("deflate", 1), ("gzip", 6), ("zlib", 9)):
        c = api.Compressor(lvl)
        d = api.Decompressor()
        bound = (c.bound(fmt, size) + 15) // 16 * 16
        in_off = torch.arange(n, dtype=torch.int64, device="cuda") * size
        in_n = torch.full((n,), size, dtype=torch.int64, device="cuda")
        comp = torch.zeros(n * bound, dtype=torch.uint8, device="cuda")
        c_off = torch.arange(n, dtype=torch.int64, device="cuda") * bound
        c_av = torch.full((n,), bound, dtype=torch.int64, device="cuda")
        c_n = torch.zeros(n, dtype=torch.int64, device="cuda")
        c.compress_batch(fmt, data, in_off, in_n, comp, c_off, c_av, c_n)
        torch.cuda.synchronize()
        sizes = c_n.cpu().numpy()
        assert (sizes > 0).all() and (sizes <= c.bound(fmt, size)).all()
        out = torch.zeros(n * size, dtype=torch.uint8, device="cuda")
        res = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        d.decompress_batch(fmt, comp, c_off, c_n, out, in_off, in_n, res)
        torch.cuda.synchronize()
        assert res.cpu().numpy().tolist() == [0] * n
        assert torch.equal(out, data)
        # and the oracle agrees on a few of them
        cb = comp.cpu().numpy()
        for i in (0, 5, 7, n - 1):
            z = cb[i * bound:i * bound + sizes[i]].tobytes()
            _check_roundtrip(oracle, fmt, chunks[i], z, (fmt, i))
        c.close()
        d.close()


@pytest.mark.timeout(180)
def test_many_buffers_per_workgroup(oracle):
    """A batch several times the number of CUs: every workgroup takes buffer after buffer
    from the batch's counter, so whatever state a buffer leaves in the LDS or in the
    workgroup's HBM scratch meets the next buffer.  (A variant of the token emission that
    passed every single-buffer-per-workgroup test hung exactly here: round 5,
    tools/experiments/.)  Checked against zlib and the oracle on a sample, against our own
    decoder on every byte; under a timeout, because the failure to expect is a hang."""
    import torch
    from libdeflate_amd import api
    n, size = 2560, 65536          # 10 buffers per workgroup on 256 CUs
    chunks = datagen.batch(n, size, 0x0E110077, distinct=48)
    data = torch.frombuffer(bytearray(b"".join(chunks)), dtype=torch.uint8).cuda()
    in_off = torch.arange(n, dtype=torch.int64, device="cuda") * size
    in_n = torch.full((n,), size, dtype=torch.int64, device="cuda")
    for fmt, lvl in (("gzip", 6), ("deflate", 1), ("zlib", 9)):
        c = api.Compressor(lvl)
        d = api.Decompressor()
        bound = (c.bound(fmt, size) + 15) // 16 * 16
        comp = torch.zeros(n * bound, dtype=torch.uint8, device="cuda")
        c_off = torch.arange(n, dtype=torch.int64, device="cuda") * bound
        c_av = torch.full((n,), bound, dtype=torch.int64, device="cuda")
        c_n = torch.zeros(n, dtype=torch.int64, device="cuda")
        c.compress_batch(fmt, data, in_off, in_n, comp, c_off, c_av, c_n)
        torch.cuda.synchronize()
        sizes = c_n.cpu().numpy()
        assert (sizes > 0).all() and (sizes <= c.bound(fmt, size)).all()
        out = torch.zeros(n * size, dtype=torch.uint8, device="cuda")
        res = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        d.decompress_batch(fmt, comp, c_off, c_n, out, in_off, in_n, res)
        torch.cuda.synchronize()
        assert int((res != 0).sum()) == 0
        assert torch.equal(out, data)
        cb = comp.cpu().numpy()
        for i in range(7, n, 211):
            z = cb[i * bound:i * bound + sizes[i]].tobytes()
            _check_roundtrip(oracle, fmt, chunks[i], z, (fmt, i))
        # the same chunk gives the same stream wherever it sits in the batch
        first = {}
        for i in range(n):
            k = i % 48
            z = cb[i * bound:i * bound + sizes[i]].tobytes()
            assert first.setdefault(k, z) == z, (fmt, i)
        c.close()
        d.close()


def test_small_blocks_zlib_level9(oracle):
    """Config-5 shape, 262 144 blocks (1 GiB): 4 KiB filesystem-block mix,
    zlib, level 9, device batch round trip, every byte compared + Adler-32
    footers checked by the decoder; a sample is also checked by the oracle
    and the reference."""
    import torch
    from libdeflate_amd import api
    n, size = 262144, 4096
    chunks = [datagen.chunk(i, size, 0x0E110005, datagen.MIX4K) for i in range(256)]
    chunks = [chunks[i % 256] for i in range(n)]
    data = torch.frombuffer(bytearray(b"".join(chunks)), dtype=torch.uint8).cuda()
    c, d = api.Compressor(9), api.Decompressor()
    bound = (c.bound("zlib", size) + 15) // 16 * 16
    in_off = torch.arange(n, dtype=torch.int64, device="cuda") * size
    in_n = torch.full((n,), size, dtype=torch.int64, device="cuda")
    comp = torch.zeros(n * bound, dtype=torch.uint8, device="cuda")
    c_off = torch.arange(n, dtype=torch.int64, device="cuda") * bound
    c_av = torch.full((n,), bound, dtype=torch.int64, device="cuda")
    c_n = torch.zeros(n, dtype=torch.int64, device="cuda")
    c.compress_batch("zlib", data, in_off, in_n, comp, c_off, c_av, c_n, max_chunk=size)
    out = torch.zeros(n * size, dtype=torch.uint8, device="cuda")
    res = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    d.decompress_batch("zlib", comp, c_off, c_n, out, in_off, in_n, res)
    torch.cuda.synchronize()
    sizes = c_n.cpu().numpy()
    assert (sizes > 0).all() and (sizes <= c.bound("zlib", size)).all()
    assert int((res != 0).sum()) == 0
    assert torch.equal(out, data)
    cb = comp.cpu().numpy()
    for i in range(0, 256, 17):
        z = cb[i * bound:i * bound + sizes[i]].tobytes()
        _check_roundtrip(oracle, "zlib", chunks[i], z, ("4k", i))
    print("4 KiB zlib L9 ratio", sizes.sum() / (n * size))


@pytest.mark.parametrize("level", [1, 6, 9, 11])
def test_large_single_buffer_is_segmented(level, oracle):
    """one large buffer through the single-buffer API is
    compressed as side-by-side sub-ranges; the result is one valid stream
    (round trip, container bytes, checksum combined from the pieces), within
    compress_bound, close to the reference's size, and 0 when it cannot fit."""
    from libdeflate_amd import api
    from tests import oracle_util
    ref = oracle_util.load_ref()
    c = api.Compressor(level)
    sizes = [131072, 131073, 200000, 1 << 20, 3 * (1 << 20) + 12345]
    for i, n in enumerate(sizes):
        kinds = [0, 5, 6, 7, 1]
        d = b"".join(datagen.chunk(kinds[(i + k) % 5], 65536, 0x0E110040 + k)
                     for k in range((n + 65535) // 65536))[:n]
        for fmt in ("deflate", "zlib", "gzip"):
            z = c.compress(fmt, d)
            _check_roundtrip(oracle, fmt, d, z, ("large", level, fmt, n))
            assert len(z) <= c.bound(fmt, n)
            if ref is not None:
                zr = ref.compress(fmt, level, d)
                assert len(z) <= 1.03 * len(zr) + 64, (level, fmt, n, len(z), len(zr))
            if fmt == "gzip":
                assert z[:4] == b"\x1f\x8b\x08\x00"
                assert int.from_bytes(z[-4:], "little") == n
                assert int.from_bytes(z[-8:-4], "little") == zlib.crc32(d)
            if fmt == "zlib":
                assert int.from_bytes(z[-4:], "big") == zlib.adler32(d)
        # does not fit -> 0 (api returns None)
        assert c.compress("gzip", d, out_avail=len(z) - 1) is None
    # a long run across many segments: distances and lengths at their limits
    d = bytes(5 * 65536 + 17)
    z = c.compress("gzip", d)
    _check_roundtrip(oracle, "gzip", d, z, ("zeros", level))


def test_block_split_follows_content():
    """lib/deflate_compress.c:2092-2218 (a10): a buffer whose content changes
    is cut into blocks near the changes (the block ends in front of the first
    tile of different content); homogeneous buffers stay one block.  Size
    within 5 % of the reference on the mixed buffers, also when the changes
    do not fall on tile boundaries."""
    import os
    import sys
    from libdeflate_amd import api
    from tests import oracle_util
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools"))
    import stream_stats
    ref = oracle_util.load_ref()
    lazy_size = {}
    for lvl in (6, 10):     # level 10: the min-cost parse re-prices at every new block
        c = api.Compressor(lvl)
        for kinds, nblocks in (((0, 0), 1), ((5, 5), 1), ((0, 5), 2), ((0, 6, 5, 7), 4)):
            d = b"".join(datagen.chunk(k, 32768, 0x0E110040 + i) for i, k in enumerate(kinds))
            z = c.compress_batch_host("deflate", [d])[0]
            out, blocks = stream_stats.stats(z)
            assert out == d
            if lvl == 6:
                assert len(blocks) == nblocks, (kinds, [b["len"] for b in blocks])
                for b in blocks[:-1]:       # cut within a tile of the change
                    end = b["start"] + b["len"]
                    assert abs(end - 32768 * round(end / 32768)) <= 4096 + 258, (kinds, end)
            else:
                assert nblocks <= len(blocks) <= nblocks + 2, (kinds, len(blocks))
            if ref is not None:
                # level 10 also pays the tile granularity of the cut against a
                # reference that re-prices the whole block several times
                assert len(z) <= (1.05 if lvl == 6 else 1.06) * len(ref.compress("deflate", lvl, d))
            if lvl == 6:
                lazy_size[kinds] = len(z)
            else:       # a tile of foreign content must not make the min-cost parse lose
                assert len(z) <= lazy_size[kinds], (kinds, len(z), lazy_size[kinds])
        # changes of content at odd offsets
        d = (datagen.chunk(0, 21000, 0x0E110048) + datagen.chunk(5, 30001, 0x0E110049) +
             datagen.chunk(6, 17777, 0x0E11004A) + datagen.chunk(0, 40000, 0x0E11004B))
        z = c.compress_batch_host("deflate", [d])[0]
        out, blocks = stream_stats.stats(z)
        assert out == d and 3 <= len(blocks) <= 6, len(blocks)
        if ref is not None:
            assert len(z) <= 1.05 * len(ref.compress("deflate", lvl, d)), (lvl, len(z))
        c.close()


def test_device_batch_ragged_unaligned(oracle):
    """Device batch with ragged sizes and byte-granular (unaligned) input and
    output offsets, neighbours packed back to back: no slot may be touched
    outside [offset, offset + size), and every chunk round-trips through both
    our decoder (unaligned too) and the oracle."""
    import torch
    from libdeflate_amd import api
    rng = np.random.default_rng(0x0E110050)
    sizes = [0, 1, 17, 4095, 4096, 4097, 5000, 33333, 65536, 65537, 100001, 131071,
             200000, 3, 70000, 12345]
    chunks = [datagen.chunk(i, s, 0x0E110051) for i, s in enumerate(sizes)]
    n = len(sizes)
    for fmt, lvl in (("gzip", 6), ("deflate", 1), ("zlib", 9), ("gzip", 12)):
        c = api.Compressor(lvl)
        d = api.Decompressor()
        in_off, pos = [], 3
        blob = bytearray(b"\xAA" * 3)
        for ch in chunks:
            in_off.append(pos)
            blob += ch
            pad = int(rng.integers(0, 3))
            blob += b"\xAA" * pad
            pos += len(ch) + pad
        blob += b"\xAA" * 64
        data = torch.frombuffer(blob, dtype=torch.uint8).cuda()
        bounds = [c.bound(fmt, s) for s in sizes]
        c_off, pos = [], 5
        for b in bounds:
            c_off.append(pos)
            pos += b + int(rng.integers(0, 2))
        comp = torch.full((pos + 64,), 0x55, dtype=torch.uint8, device="cuda")
        t = lambda v: torch.tensor(v, dtype=torch.int64, device="cuda")
        c_n = torch.zeros(n, dtype=torch.int64, device="cuda")
        c.compress_batch(fmt, data, t(in_off), t(sizes), comp, t(c_off), t(bounds), c_n)
        torch.cuda.synchronize()
        got = c_n.cpu().numpy()
        cb = comp.cpu().numpy()
        for i in range(n):
            assert 0 < got[i] <= bounds[i], (fmt, i)
            z = cb[c_off[i]:c_off[i] + got[i]].tobytes()
            _check_roundtrip(oracle, fmt, chunks[i], z, (fmt, "ragged", i))
            # bytes between this slot's data and the next slot are untouched
            end = c_off[i + 1] if i + 1 < n else len(cb)
            assert (cb[c_off[i] + got[i]:end] == 0x55).all(), (fmt, "slot overrun", i)
        assert (cb[:5] == 0x55).all()
        # decode from the unaligned compressed slots into unaligned outputs
        o_off, pos = [], 7
        for s in sizes:
            o_off.append(pos)
            pos += s + int(rng.integers(0, 3))
        out = torch.full((pos + 64,), 0x33, dtype=torch.uint8, device="cuda")
        res = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        d.decompress_batch(fmt, comp, t(c_off), c_n, out, t(o_off), t(sizes), res)
        torch.cuda.synchronize()
        assert res.cpu().numpy().tolist() == [0] * n
        ob = out.cpu().numpy()
        for i in range(n):
            assert ob[o_off[i]:o_off[i] + sizes[i]].tobytes() == chunks[i], (fmt, "decode", i)
            end = o_off[i + 1] if i + 1 < n else len(ob)
            assert (ob[o_off[i] + sizes[i]:end] == 0x33).all(), (fmt, "output overrun", i)
        c.close()
        d.close()


def test_device_compaction_and_payload_gather():
    """4096 ragged compressed outputs compacted on the device
    (libdeflate_amd_compact_batch) equal the host-side concatenation, the
    offsets are the exclusive prefix sums, and shard.gather_payload hands the
    segment through unchanged (world 1; the N > 1 exchange is covered on gloo
    in tests/test_shard.py)."""
    import torch
    from libdeflate_amd import api, shard
    n, size = 4096, 65536
    chunks = datagen.batch(n, size, 0x0E110031, distinct=64)
    c = api.Compressor(6)
    bound = (c.bound("gzip", size) + 15) // 16 * 16
    data = torch.frombuffer(bytearray(b"".join(chunks)), dtype=torch.uint8).cuda()
    in_off = torch.arange(n, dtype=torch.int64, device="cuda") * size
    in_n = torch.full((n,), size, dtype=torch.int64, device="cuda")
    comp = torch.zeros(n * bound, dtype=torch.uint8, device="cuda")
    c_off = torch.arange(n, dtype=torch.int64, device="cuda") * bound
    c_av = torch.full((n,), bound, dtype=torch.int64, device="cuda")
    c_n = torch.zeros(n, dtype=torch.int64, device="cuda")
    c.compress_batch("gzip", data, in_off, in_n, comp, c_off, c_av, c_n)
    packed, off = api.compact_batch(comp, c_off, c_n)
    torch.cuda.synchronize()
    sizes = c_n.cpu().numpy()
    offs = off.cpu().numpy()
    assert (offs[:-1] == np.concatenate([[0], np.cumsum(sizes)[:-1]])).all()
    assert offs[-1] == sizes.sum()
    host = comp.cpu().numpy()
    want = np.concatenate([host[i * bound:i * bound + sizes[i]] for i in range(n)])
    got = shard.compact(comp, c_off, c_n)
    assert got.numel() == want.size and (got.cpu().numpy() == want).all()
    assert (packed[:int(offs[-1])].cpu().numpy() == want).all()
    segs = shard.gather_payload(got, None, 1)
    assert len(segs) == 1 and torch.equal(segs[0], got)
    # odd sizes and alignments, zero-length chunks, more chunks than one scan block
    rng = np.random.default_rng(5)
    m = 5000
    lens = rng.integers(0, 200, size=m)
    lens[::7] = 0
    starts = np.cumsum(np.concatenate([[3], lens[:-1] + rng.integers(0, 9, size=m - 1)]))
    blob = rng.integers(0, 256, size=int(starts[-1] + lens[-1] + 16), dtype=np.uint8)
    d_blob = torch.from_numpy(blob).cuda()
    p2, o2 = api.compact_batch(d_blob, torch.from_numpy(starts).cuda(),
                               torch.from_numpy(lens).cuda())
    torch.cuda.synchronize()
    want2 = np.concatenate([blob[s:s + l] for s, l in zip(starts, lens)])
    assert int(o2[-1]) == want2.size
    assert (p2[:want2.size].cpu().numpy() == want2).all()


def test_batch_host_many_small_chunks():
    """65 536 x 4 KiB through the host-pointer batch entry points (packed
    pinned staging, device compaction): every stream decodes to its chunk."""
    from libdeflate_amd import api
    n, size = 65536, 4096
    chunks = datagen.batch(n, size, 0x0E110032, mix=datagen.MIX4K, distinct=512)
    c, d = api.Compressor(6), api.Decompressor()
    comp = c.compress_batch_host("zlib", chunks)
    assert all(z is not None for z in comp)
    for i in range(0, n, 997):
        assert zlib.decompress(comp[i]) == chunks[i]
    back = d.decompress_batch_host("zlib", comp, [size] * n)
    assert all(r[0] == 0 for r in back)
    assert all(back[i][3] == chunks[i] for i in range(0, n, 61))


@pytest.mark.parametrize("level", [0, 1, 3, 6, 9])
def test_small_buffer_kernel(level, oracle):
    """Batches whose chunks are all <= 4096 bytes run on the 256-thread kernel
    (deflate_small.hip; the host-pointer batch knows the sizes and picks it):
    every size class around its limits, all content kinds, all formats; and
    the device entry point with a size bound that a chunk violates reports 0
    for that chunk only."""
    import torch
    from libdeflate_amd import api
    rng = np.random.default_rng(0x0E110070 + level)
    sizes = [0, 1, 2, 3, 4, 5, 18, 19, 31, 32, 33, 51, 52, 63, 64, 65, 255, 256,
             511, 512, 513, 1000, 2047, 2048, 2049, 4000, 4094, 4095, 4096]
    sizes += [int(rng.integers(0, 4097)) for _ in range(60)]
    chunks = [_weird_chunk(rng, n) if i % 2 else datagen.chunk(i, n, 0x0E110071, datagen.MIX4K)
              for i, n in enumerate(sizes)]
    c = api.Compressor(level)
    for fmt in ("deflate", "zlib", "gzip"):
        comps = c.compress_batch_host(fmt, chunks)
        for d, z in zip(chunks, comps):
            _check_roundtrip(oracle, fmt, d, z, ("small", level, fmt, len(d)))
            assert len(z) <= c.bound(fmt, len(d))
    # too small an output slot -> 0 for that chunk, the others unaffected
    avail = [c.bound("zlib", len(d)) for d in chunks]
    big = max(range(len(chunks)), key=lambda i: len(chunks[i]))
    avail[big] = 20
    comps = c.compress_batch_host("zlib", chunks, out_avail=avail)
    assert comps[big] is None and all(z is not None for i, z in enumerate(comps) if i != big)
    # a chunk above the stated bound: 0 for it alone
    n = 8
    data = [datagen.chunk(i, 4096, 0x0E110072, datagen.MIX4K) for i in range(n)]
    blob = torch.frombuffer(bytearray(b"".join(data) + bytes(4096)), dtype=torch.uint8).cuda()
    in_off = torch.arange(n, dtype=torch.int64, device="cuda") * 4096
    in_n = torch.full((n,), 4096, dtype=torch.int64, device="cuda")
    in_n[3] = 5000
    bound = 5120
    out = torch.zeros(n * bound, dtype=torch.uint8, device="cuda")
    o_off = torch.arange(n, dtype=torch.int64, device="cuda") * bound
    o_av = torch.full((n,), bound, dtype=torch.int64, device="cuda")
    o_n = torch.full((n,), -1, dtype=torch.int64, device="cuda")
    c.compress_batch("deflate", blob, in_off, in_n, out, o_off, o_av, o_n, max_chunk=4096)
    torch.cuda.synchronize()
    got = o_n.cpu().numpy()
    assert got[3] == 0 and (np.delete(got, 3) > 0).all()
    c.close()


def test_small_buffer_of_two_kinds(oracle):
    """The small-buffer kernel takes a 4 KiB buffer as two tiles of 2048
    positions (deflate_small.hip): a buffer whose halves are of different
    content may end its first block between them (the split rule of
    lib/deflate_compress.c:2092-2218 once per tile), the round trip holds for
    every pair of kinds and every cut position around the tile boundary, and
    the batch is within 3 % of the reference's size at the same level."""
    from libdeflate_amd import api
    from tests import oracle_util
    ref = oracle_util.load_ref()
    bufs = []
    for i, (ka, kb) in enumerate(((0, 7), (7, 0), (0, 5), (5, 6), (6, 0), (2, 7), (0, 0), (7, 7))):
        for cut in (1, 1000, 2046, 2047, 2048, 2049, 2050, 2306, 3000, 4095):
            bufs.append(datagen.chunk(ka, cut, 0x0E110090 + i, datagen.MIX4K) +
                        datagen.chunk(kb, 4096 - cut, 0x0E1100A0 + i, datagen.MIX4K))
    for level in (1, 6, 9):
        c = api.Compressor(level)
        comps = c.compress_batch_host("zlib", bufs)
        for d, z in zip(bufs, comps):
            _check_roundtrip(oracle, "zlib", d, z, ("two kinds", level))
            assert len(z) <= c.bound("zlib", len(d))
        if ref is not None:
            ours = sum(map(len, comps))
            theirs = sum(len(ref.compress("zlib", level, d)) for d in bufs)
            print(f"4 KiB buffers of two kinds, level {level}: {ours} bytes, reference {theirs}")
            assert ours <= 1.03 * theirs, (level, ours, theirs)
        c.close()


def test_first_call_small_batch_then_large():
    """A process whose FIRST compress call is a batch of small buffers (the
    256-thread kernel) must still be able to launch the 1024-thread kernels
    afterwards: the per-device LDS limits of all three kernels are set on the
    first call, each to its own size.  Needs a fresh process."""
    import subprocess
    import sys
    code = r'''
import sys, zlib
sys.path.insert(0, %r)
from tests import datagen
from libdeflate_amd import api
c = api.Compressor(6)
small = [datagen.chunk(i, 3000, 0x0E110080, datagen.MIX4K) for i in range(16)]
for d, z in zip(small, c.compress_batch_host("zlib", small)):
    assert zlib.decompress(z) == d
big = [datagen.chunk(i, 65536, 0x0E110081) for i in range(8)]
for d, z in zip(big, c.compress_batch_host("zlib", big)):
    assert z is not None and zlib.decompress(z) == d
c12 = api.Compressor(12)
for d, z in zip(big[:2], c12.compress_batch_host("zlib", big[:2])):
    assert z is not None and zlib.decompress(z) == d
print("ok")
''' % os.path.join(os.path.dirname(__file__), "..")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-2000:])


def test_input_over_4gib(ref):
    """An input of 4.5 GiB through the single-buffer API (the reference takes
    any size_t, lib/deflate_compress.c:4030-4072): cut into 64 KiB segments,
    one stream, ISIZE mod 2^32, CRC-32 combined over 73 728 pieces; decoded by
    the REAL reference and compared byte for byte."""
    import ctypes
    import psutil
    from libdeflate_amd import api
    n = (9 << 29) + 12345                # 4.5 GiB + an odd tail
    if psutil.virtual_memory().available < 4 * n:
        pytest.skip("needs ~18 GiB of free host memory")
    block = np.frombuffer(b"".join(datagen.chunk(i, 1 << 20, 0x0E110077) for i in range(16)),
                          dtype=np.uint8)
    data = np.tile(block, n // block.size + 1)[:n]
    # make every 16 MiB repetition differ (no trivially repeating stream)
    data[::1 << 24] = (np.arange((n + (1 << 24) - 1) >> 24) & 255).astype(np.uint8)
    c = api.Compressor(1)
    bound = c.bound("gzip", n)
    out = np.empty(bound, dtype=np.uint8)
    r = c._lib.libdeflate_gzip_compress(c._h, data.ctypes.data_as(ctypes.c_void_p), n,
                                        out.ctypes.data_as(ctypes.c_void_p), bound)
    c.close()
    assert 0 < r <= bound
    # ISIZE is n mod 2^32 (gzip_compress.c:78-79)
    assert int.from_bytes(out[r - 4:r].tobytes(), "little") == n & 0xFFFFFFFF
    back = np.empty(n, dtype=np.uint8)
    ai, ao = ctypes.c_size_t(0), ctypes.c_size_t(0)
    res = ref.lib.libdeflate_gzip_decompress_ex(
        ref._d, ctypes.cast(out.ctypes.data, ctypes.c_char_p), r,
        back.ctypes.data_as(ctypes.c_void_p), n, ctypes.byref(ai), ctypes.byref(ao))
    assert (res, ai.value, ao.value) == (0, r, n)
    assert np.array_equal(back, data)
    # and back through this library: a stream of more than 4 GiB is decoded by
    # the many-wave path (the one-wave kernels index a stream with 32 bits)
    from libdeflate_amd import binding
    back[:] = 0
    d = api.Decompressor()
    ai, ao = ctypes.c_size_t(0), ctypes.c_size_t(0)
    res = d._lib.libdeflate_gzip_decompress_ex(
        d._h, out.ctypes.data_as(ctypes.c_void_p), r,
        back.ctypes.data_as(ctypes.c_void_p), n, ctypes.byref(ai), ctypes.byref(ao))
    st = binding.stream_stats()
    d.close()
    assert (res, ai.value, ao.value) == (0, r, n), st
    assert st["parallel"] == 1, st
    assert np.array_equal(back, data)


@pytest.mark.parametrize("level", [1, 6, 9, 12])
def test_output_is_deterministic(level):
    """The compress kernel runs several stages of two tiles side by side and
    hands work to whichever wave is free; none of that may show in the bytes:
    the same batch compressed three times (twice by one object, once by
    another, with a different batch in between) gives identical streams."""
    from libdeflate_amd import api
    rng = np.random.default_rng(0x0DE7 + level)
    chunks = datagen.batch(96, 65536, 0x0E110031, distinct=96)
    chunks += [_weird_chunk(rng, int(rng.integers(1, 150000))) for _ in range(32)]
    other = datagen.batch(64, 40000, 0x0E110032, distinct=64)
    c1, c2 = api.Compressor(level), api.Compressor(level)
    a = c1.compress_batch_host("gzip", chunks)
    c1.compress_batch_host("gzip", other)
    b = c1.compress_batch_host("gzip", chunks)
    c = c2.compress_batch_host("gzip", chunks)
    c1.close()
    c2.close()
    assert a == b == c


def test_streams_are_the_recorded_ones():
    """The bytes this library produces are not pinned by the reference
    (libdeflate.h:76-83), but they are a pure function of the input and the
    build: `tools/digest_deflate.py --quick` (a fixed seeded input set at levels
    1, 6, 9, 12 through both compress kernels and the segmented path, every
    stream decoded by zlib) must print the digests recorded in
    tests/golden/digest_quick.txt for this build.  A change of a kernel's
    schedule keeps them; a change of policy regenerates the file (and says in
    DESIGN.md what it did to the sizes)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "digest_deflate.py"), "--quick"],
                       capture_output=True, text=True, timeout=300, cwd=root)
    got = [l for l in r.stdout.splitlines() if l.startswith("L")]
    want = open(os.path.join(root, "tests", "golden", "digest_quick.txt")).read().splitlines()
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert got == want, (
        "the compressed bytes of the fixed input set changed.  If that is a deliberate "
        "change of policy (every stream was still decoded by zlib: see above), regenerate the "
        "fixture - `python tools/digest_deflate.py --quick | grep ^L > tests/golden/digest_quick.txt` "
        "on a GPU box - and say in DESIGN.md what it did to the sizes; a change of schedule "
        "must not get here.  Lines that differ:\n" + "\n".join(l for l in got if l not in want))
