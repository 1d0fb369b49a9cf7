"""ONE large stream on many waves (csrc/inflate_stream.hip, host_stream.hip)
through the reference's single-buffer calls, libdeflate_*_decompress[_ex]
(libdeflate.h:242-315) - what programs/gzip.c:187-303 and the 1 MiB chunks of
programs/benchmark.c:543-544 hand to the library.  Streams compressed by the
real reference (zlib where oracle/_ref did not travel); every byte, actual_in /
actual_out and every result code against the oracle, and
libdeflate_amd_stream_stats() says which path answered."""
import os
import time
import zlib

import numpy as np
import pytest

from libdeflate_amd import binding
from tests import datagen, oracle_util, streams

pytestmark = pytest.mark.gpu
WBITS = {"deflate": -15, "zlib": 15, "gzip": 31}


@pytest.fixture(scope="module")
def dec():
    from libdeflate_amd import api
    d = api.Decompressor()
    yield d
    d.close()


@pytest.fixture(scope="module")
def comp():
    ref = oracle_util.load_ref()
    if ref is not None:
        return lambda fmt, lvl, d: ref.compress(fmt, lvl, d)
    return lambda fmt, lvl, d: streams._zcompress(fmt, min(lvl, 9), d)


def _data(kind, n, seed):
    if kind == "text":
        return datagen.text_chunk(n, seed)
    return b"".join(datagen.chunk(i, 65536, seed) for i in range((n + 65535) // 65536))[:n]


@pytest.mark.parametrize("mib,level,kind", [
    (1, 1, "text"), (1, 6, "mix"), (1, 12, "text"),
    (4, 1, "mix"), (4, 6, "text"), (4, 12, "mix"),
    (16, 1, "text"), (16, 6, "text"), (16, 6, "mix"), (16, 12, "text")])
def test_large_streams(dec, comp, mib, level, kind):
    n = mib << 20
    data = _data(kind, n, 0x51000 + mib + level)
    fmt = ("gzip", "zlib", "deflate")[(mib + level) % 3]
    z = comp(fmt, level, data)
    r = dec.decompress_ex(fmt, z, n)
    st = binding.stream_stats()
    assert r[:3] == (0, len(z), n), (r[:3], st)
    assert r[3] == data
    assert st["parallel"] == 1, st
    assert st["bytes"] == n
    # exact fill (actual_out_nbytes_ret = NULL) and trailing bytes
    r2 = dec.decompress(fmt, z + b"\x00\x01\x02", n, False)
    assert r2[0] == 0 and r2[2] == data
    assert binding.stream_stats()["parallel"] == 1
    print(f"{mib} MiB {kind} L{level} {fmt}: {st}")


def test_sixteen_mib_rate(dec, comp):
    """a 16 MiB reference-compressed gzip stream through
    libdeflate_gzip_decompress, host to host, at one core's rate or better."""
    n = 16 << 20
    data = datagen.text_chunk(n, 0x51777)
    z = comp("gzip", 6, data)
    out = np.zeros(n, dtype=np.uint8)
    from ctypes import c_size_t, c_void_p, byref
    lib = binding.load()
    zin = np.frombuffer(z, dtype=np.uint8)
    best = 1e9
    for _ in range(5):
        ao = c_size_t(0)
        t0 = time.perf_counter()
        r = lib.libdeflate_gzip_decompress(dec._h, zin.ctypes.data_as(c_void_p), zin.size,
                                           out.ctypes.data_as(c_void_p), n, byref(ao))
        best = min(best, time.perf_counter() - t0)
        assert r == 0 and ao.value == n
    assert out.tobytes() == data
    st = binding.stream_stats()
    assert st["parallel"] == 1, st
    print(f"16 MiB gzip L6 text, host to host: {best * 1e3:.2f} ms = {n / best / 1e9:.2f} GB/s; {st}")
    assert n / best / 1e9 >= 1.2


def test_result_codes_of_damaged_large_streams(dec, comp, oracle):
    n = 3 << 20
    data = _data("mix", n, 0x52000)
    for fmt in ("deflate", "gzip", "zlib"):
        z = comp(fmt, 6, data)
        rng = np.random.default_rng(7)
        variants = [("ok", z, n), ("short", z, n + 1), ("nospace", z, n - 1),
                    ("trunc1", z[:len(z) // 3], n), ("trunc2", z[:-9], n),
                    ("trunc3", z[:-1], n)]
        for k in range(6):
            b = bytearray(z)
            pos = int(rng.integers(0, len(b)))
            b[pos] ^= 1 << int(rng.integers(0, 8))
            variants.append((f"flip{k}@{pos}", bytes(b), n))
        b = bytearray(z)
        b[-5] ^= 0x40       # footer (or, raw: the stream's last bytes)
        variants.append(("footer", bytes(b), n))
        for name, s, avail in variants:
            for want in (True, False):
                got = dec.decompress_ex(fmt, s, avail, want)
                exp = oracle.decompress_ex(fmt, s, avail, want)
                assert got[0] == exp[0], (fmt, name, want, got[:3], exp[:3], binding.stream_stats())
                if exp[0] == 0:
                    assert got[1] == exp[1] and got[3] == exp[3], (fmt, name, want)
                    if want:
                        assert got[2] == exp[2]


def _static_block_stream(data):
    """ONE static Huffman block holding `data` as literals."""
    w = streams.BitWriter()
    w.put(1, 1)
    w.put(1, 2)
    for b in data:
        streams._static_lit(w, b)
    streams._static_lit(w, 256)
    return w.finish()


def test_streams_the_finder_cannot_enter(dec, oracle):
    """Stored-only, static-only and one-giant-block streams have no dynamic
    block header to find (or only one).  Round 5: they are decoded on many
    waves all the same - the host walks runs of stored blocks itself (one
    chunk per block, no count pass), and a static block needs no header to be
    entered anywhere (chunks under the static codes); the bytes and codes are
    the oracle's."""
    rnd = datagen.random_chunk(2 << 20, 5)
    stored = streams._zcompress("deflate", 0, rnd)
    r = dec.decompress_ex("deflate", stored, len(rnd))
    st = binding.stream_stats()
    print("stored-only:", st)
    assert r == (0, len(stored), len(rnd), rnd)
    assert st["parallel"] == 1 and st["chunks_decoded"] >= 32, st
    # truncated inside a stored block, and a stored LEN / NLEN that disagree
    # truncated inside a block; the third block's NLEN no longer matches
    h3 = 0
    for _ in range(2):
        h3 += 5 + (stored[h3 + 1] | stored[h3 + 2] << 8)
    for bad in (stored[:len(stored) - 70000],
                stored[:h3 + 3] + bytes([stored[h3 + 3] ^ 0x40]) + stored[h3 + 4:]):
        got = dec.decompress_ex("deflate", bad, len(rnd))
        assert got[0] == oracle.decompress_ex("deflate", bad, len(rnd))[0] != 0
    txt = datagen.text_chunk(1 << 20, 9)
    co = zlib.compressobj(6, zlib.DEFLATED, -15, 9, zlib.Z_FIXED)
    fixed = co.compress(txt) + co.flush()
    r = dec.decompress_ex("deflate", fixed, len(txt))
    st = binding.stream_stats()
    print("static blocks:", st)
    assert r == (0, len(fixed), len(txt), txt)
    assert st["parallel"] == 1 and st["chunks_decoded"] >= 32, st
    for cut in (len(fixed) // 2, len(fixed) - 1):
        got = dec.decompress_ex("deflate", fixed[:cut], len(txt))
        assert got[0] == oracle.decompress_ex("deflate", fixed[:cut], len(txt))[0] != 0
    giant = _static_block_stream(txt[:200000])
    r = dec.decompress_ex("deflate", giant, 200000)
    st = binding.stream_stats()
    print("one static block:", st)
    assert r == (0, len(giant), 200000, txt[:200000])
    assert st["parallel"] == 1 and st["chunks_decoded"] >= 8, st
    # stored blocks between Huffman blocks (zlib's full flushes leave empty ones)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    mixed = b""
    for i in range(24):
        mixed += co.compress(txt[i * 40000:(i + 1) * 40000]) + co.flush(zlib.Z_FULL_FLUSH)
    mixed += co.flush()
    r = dec.decompress_ex("deflate", mixed, 24 * 40000)
    assert r == (0, len(mixed), 24 * 40000, txt[:24 * 40000]), binding.stream_stats()
    assert binding.stream_stats()["parallel"] == 1
    # one dynamic block as large as zlib makes them, then the same bytes twice
    co = zlib.compressobj(9, zlib.DEFLATED, -15, 9)
    big = datagen.lowentropy_chunk(600000, 3)
    z = co.compress(big) + co.flush()
    r = dec.decompress_ex("deflate", z, len(big))
    print("zlib memLevel 9:", binding.stream_stats())
    assert r == (0, len(z), len(big), big)
    zeros = bytes(8 << 20)
    z = streams._zcompress("gzip", 9, zeros)
    r = dec.decompress_ex("gzip", z, len(zeros))
    print("8 MiB of zeros:", binding.stream_stats())
    assert r == (0, len(z), len(zeros), zeros)


def test_blocks_of_one_codeword_length(dec, comp, oracle):
    """Dynamic blocks over incompressible bytes - literal codewords of (nearly)
    one length - inside a large stream: a parse started at a wrong bit never
    falls in step there, so warm-ups fail.  The host reads such a block's
    header itself and plans its inner chunks at EXACT starts, one per bit a
    literal across the planned position can end at (host_stream.hip,
    one_length_code()); the chain picks the true one.  Round 6: the blocks of
    one length between text, alone, with matches across the planned starts, and
    damaged - bytes and codes are the oracle's, and the chain closes in few
    rounds (repairs stay far below one per chunk)."""
    rng = np.random.default_rng(0x1E6)
    txt = datagen.text_chunk(3 << 20, 41)
    parts = []
    for i in range(12):
        parts.append(txt[i * 200000:(i + 1) * 200000])
        parts.append(rng.integers(0, 256, 70000, dtype=np.uint8).tobytes())
    mixed = b"".join(parts)
    z = comp("gzip", 6, mixed)
    r = dec.decompress_ex("gzip", z, len(mixed))
    st = binding.stream_stats()
    print("text and incompressible bytes:", st)
    assert r == (0, len(z), len(mixed), mixed), (r[:3], st)
    assert st["parallel"] == 1, st
    # Huffman-only blocks over bytes of 256 / 250 / 64 values (8 bits; 7 and 8; 6)
    for nv in (256, 250, 64):
        raw = rng.integers(0, nv, 3 << 20, dtype=np.uint8).tobytes()
        co = zlib.compressobj(6, zlib.DEFLATED, -15, 9, zlib.Z_HUFFMAN_ONLY)
        h = co.compress(raw) + co.flush()
        r = dec.decompress_ex("deflate", h, len(raw))
        st = binding.stream_stats()
        print(f"Huffman-only, {nv} values:", st)
        assert r == (0, len(h), len(raw), raw), (nv, r[:3], st)
        assert st["parallel"] == 1, st
        assert st["repairs"] <= st["chunks_decoded"], st
        # a damaged byte in the middle: whatever the oracle says
        bad = h[:len(h) // 2] + bytes([h[len(h) // 2] ^ 0x10]) + h[len(h) // 2 + 1:]
        got = dec.decompress_ex("deflate", bad, len(raw))
        exp = oracle.decompress_ex("deflate", bad, len(raw))
        assert got[0] == exp[0], (nv, got[:3], exp[:3], binding.stream_stats())
        if exp[0] == 0:
            assert got[1:] == exp[1:]
    # incompressible bytes with a repeated stretch now and then: matches (long
    # tokens) across planned starts, which no exact start covers - repairs
    blocks = []
    for i in range(40):
        b = rng.integers(0, 256, 60000, dtype=np.uint8).tobytes()
        blocks.append(b + b[1000:1000 + 300] * 3)
    rep = b"".join(blocks)
    z = comp("zlib", 6, rep)
    r = dec.decompress_ex("zlib", z, len(rep))
    st = binding.stream_stats()
    print("incompressible with repeats:", st)
    assert r == (0, len(z), len(rep), rep), (r[:3], st)


def test_small_chunks_everything_through_the_stream_path(dec, oracle, comp, monkeypatch):
    """Every stream, however small, through the many-wave path with 4 KiB
    chunks: the chunk kernels' edge cases (a stream shorter than a round, the
    last bytes of the input, empty blocks, tiny blocks) against the oracle."""
    monkeypatch.setenv("LDA_STREAM_PAR_MIN", "0")
    monkeypatch.setenv("LDA_STREAM_CHUNK", "4096")
    binding.reload_env()
    cases = streams.random_cases(31, 120, compress=comp,
                                 sizes=[0, 1, 5, 100, 1000, 5000, 20000, 70000, 300000])
    cases += [(f, s, a, w, t) for f, s, a, w, t in streams.garbage_cases(32, 60)]
    for s, want in streams.stored_then_match_streams():
        cases.append(("deflate", s, len(want), True, "stored_then_match"))
    for name, s, want in streams.parallel_round_streams():
        cases.append(("deflate", s, len(want), True, name))
    for s in (streams.empty_static_blocks(), streams.empty_dynamic_blocks()):
        cases.append(("deflate", s, 10000, True, "slow"))
    npar = 0
    for fmt, s, avail, want, tag in cases:
        got = dec.decompress_ex(fmt, s, avail, want)
        npar += binding.stream_stats()["parallel"]
        exp = oracle.decompress_ex(fmt, s, avail, want)
        assert got[0] == exp[0], (tag, fmt, got[:3], exp[:3], binding.stream_stats())
        if exp[0] == 0:
            assert got[1] == exp[1] and got[3] == exp[3], (tag, fmt, binding.stream_stats())
            if want:
                assert got[2] == exp[2]
    print(f"{npar} of {len(cases)} cases were answered by the many-wave path")
    assert npar >= len(cases) // 8


def test_switch_off(dec, comp, monkeypatch):
    data = datagen.text_chunk(1 << 20, 77)
    z = comp("gzip", 6, data)
    monkeypatch.setenv("LDA_NO_STREAM_PAR", "1")
    binding.reload_env()
    assert dec.decompress_ex("gzip", z, len(data)) == (0, len(z), len(data), data)
    assert binding.stream_stats()["parallel"] == 0


def test_input_windows(dec, comp, oracle, monkeypatch):
    """The input is taken in windows (4 MiB, then 4x as much each time): a
    stream longer than the first window carries its state - inside a block or
    at a block boundary - into the next one; a stream that ends inside the
    first window leaves the rest of the buffer alone.  Small windows here, so
    that a few MiB cross many of them."""
    data = _data("mix", 3 << 20, 0x53000)
    for win in ("32768", "65536", "262144"):
        monkeypatch.setenv("LDA_STREAM_WINDOW", win)
        monkeypatch.setenv("LDA_STREAM_PAR_MIN", "0")
        binding.reload_env()
        for fmt, lvl in (("gzip", 6), ("deflate", 1), ("zlib", 12), ("deflate", 0)):
            z = comp(fmt, lvl, data)
            r = dec.decompress_ex(fmt, z, len(data))
            st = binding.stream_stats()
            assert r == (0, len(z), len(data), data), (win, fmt, lvl, r[:3], st)
            assert st["parallel"] == 1, (win, fmt, lvl, st)
        # static-only stream: no candidates anywhere, window after window
        co = zlib.compressobj(6, zlib.DEFLATED, -15, 9, zlib.Z_FIXED)
        txt = data[:500000]
        z = co.compress(txt) + co.flush()
        assert dec.decompress_ex("deflate", z, len(txt)) == (0, len(z), len(txt), txt)
        # truncated: the final block is never seen -> the oracle's verdict
        z = comp("gzip", 6, data)
        for cut in (len(z) // 2, len(z) - 9):
            got = dec.decompress_ex("gzip", z[:cut], len(data))
            assert got[0] == oracle.decompress_ex("gzip", z[:cut], len(data))[0]


def test_stored_run_into_the_footer(dec, oracle, monkeypatch):
    """A partial input window that ends INSIDE the gzip footer, and a final
    stored block whose LEN reaches into that footer: the reference hands its
    decoder in_nbytes - header - footer bytes and says BAD_DATA; the many-wave
    path must not count the block as the stream's end (nor read the footer
    behind the caller's buffer) - the oracle's code either way.  The honest
    stream of the same shape decodes."""
    import struct
    monkeypatch.setenv("LDA_STREAM_WINDOW", "32768")
    monkeypatch.setenv("LDA_STREAM_PAR_MIN", "0")
    binding.reload_env()
    rng = np.random.default_rng(0xF007)
    for t in range(1, 8):
        for over in (0, 1, t, 8):
            raw_len = 32768 - t         # the window ends t bytes into the footer
            payload, raw = bytearray(), bytearray()
            while raw_len - len(raw) > 5 + 1000:
                blk = rng.integers(0, 256, 1000, dtype=np.uint8).tobytes()
                raw += b"\x00" + struct.pack("<HH", 1000, 1000 ^ 0xFFFF) + blk
                payload += blk
            k = raw_len - len(raw) - 5
            blk = rng.integers(0, 256, k, dtype=np.uint8).tobytes()
            raw += b"\x01" + struct.pack("<HH", k + over, (k + over) ^ 0xFFFF) + blk
            payload += blk
            assert len(raw) == raw_len
            z = (b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\xff" + bytes(raw) +
                 struct.pack("<II", zlib.crc32(bytes(payload)), len(payload)))
            got = dec.decompress_ex("gzip", z, len(payload) + 64)
            exp = oracle.decompress_ex("gzip", z, len(payload) + 64)
            assert got[0] == exp[0], (t, over, got[:3], exp[:3], binding.stream_stats())
            if over == 0:
                assert exp[0] == 0 and got[1:] == (len(z), len(payload), bytes(payload))


def test_members_one_after_the_other(dec, comp):
    """programs/gzip.c:236-299: the caller hands the rest of the file to every
    call; a member that ends inside the first window must not make the library
    copy and search everything behind it (here: 40 members of 1 MiB each, every
    call sees what is left of the 13 MB file)."""
    members = [datagen.text_chunk(1 << 20, 0x54000 + i) for i in range(40)]
    z = b"".join(comp("gzip", 6, m) for m in members)
    t0 = time.perf_counter()
    r = dec.gzip_decompress_members(z, 40 << 20)
    dt = time.perf_counter() - t0
    assert r[0] == 0 and r[1] == len(z) and r[3] == 40
    assert r[4] == b"".join(members)
    print(f"40 gzip members of 1 MiB, member after member: {dt * 1e3:.1f} ms "
          f"({(40 << 20) / dt / 1e9:.2f} GB/s)")
    assert dt < 2.0


def test_block_finder_against_the_block_map(dec, comp, oracle):
    """The finder (every bit offset tried as a dynamic block header) against
    the stream's real block structure (oracle_deflate_block_map): it finds
    the dynamic blocks - all but, at most, one in fifty - and on compressed
    text nothing else; among incompressible bytes (stored blocks) it may take
    a few header-like patterns for block starts, which the chain rejects."""
    for kind, level, max_false in (("text", 6, 2), ("text", 1, 2), ("mix", 6, 400)):
        data = _data(kind, 4 << 20, 0x55000 + level)
        z = comp("deflate", level, data)
        blocks, r = oracle.block_map(z, len(data))
        assert r == 0
        dyn = sum(1 for b in blocks if b[2] == 2)
        got = dec.decompress_ex("deflate", z, len(data))
        st = binding.stream_stats()
        assert got == (0, len(z), len(data), data)
        assert st["parallel"] == 1 and st["windows"] == 1, st
        print(f"{kind} L{level}: {len(blocks)} blocks, {dyn} dynamic; finder: "
              f"{st['blocks_found']} (first filter {st['filter_a']}), repairs {st['repairs']}")
        assert st["blocks_found"] >= dyn - max(1, dyn // 50), (dyn, st)
        assert st["blocks_found"] <= dyn + max_false, (dyn, st)
