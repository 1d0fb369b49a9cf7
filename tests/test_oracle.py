"""Pins the oracle (CPU restatement) before anything trusts it:
 (a) the reference's own unit-test vectors, rebuilt in tests/streams.py;
 (b) the committed golden fixtures produced by the real reference;
 (c) live against oracle/_ref (the reference compiled from its sources) when
     it is present - randomized valid / truncated / corrupted streams.
Runs on CPU (`-m "not gpu"`)."""
import json
import os
import zlib

import pytest

from tests import datagen, streams

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_checksums_vs_zlib(oracle):
    import random
    rng = random.Random(3)
    for n in [0, 1, 2, 3, 15, 16, 17, 255, 256, 4999, 5552, 5553, 70000]:
        d = bytes(rng.randrange(256) for _ in range(n))
        for init in (None, rng.getrandbits(32)):
            c0 = 0 if init is None else init
            a0 = 1 if init is None else ((init % 65521) << 16) | (init >> 16) % 65521
            assert oracle.crc32(d, c0) == zlib.crc32(d, c0)
            assert oracle.adler32(d, a0) == zlib.adler32(d, a0)
    # Adler overflow vector: programs/test_checksums.c:184-196
    d = b"\xff" * 5553
    init = (65520 << 16) | 65520
    assert oracle.adler32(d, init) == zlib.adler32(d, init)
    # multipart == one shot: test_checksums.c:73-84
    d = datagen.text_chunk(10000, 1)
    assert oracle.crc32(d[5000:], oracle.crc32(d[:5000])) == oracle.crc32(d)
    assert oracle.adler32(d[777:], oracle.adler32(d[:777])) == oracle.adler32(d)
    # NULL buffer rules are exercised through ctypes None
    assert oracle.lib.oracle_crc32(1234, None, 1234) == 0
    assert oracle.lib.oracle_adler32(1234, None, 1234) == 1


def test_reference_unit_test_vectors(oracle):
    s, want = streams.incomplete_empty_offset_code()
    assert oracle.decompress_ex("deflate", s, 4)[0::3] == (0, want)
    assert zlib.decompress(s, -15) == want
    s, want = streams.incomplete_singleton_litlen()
    r = oracle.decompress_ex("deflate", s, 0)
    assert r[0] == 0 and r[2] == 0
    for nz in (False, True):
        s, want = streams.incomplete_singleton_offset(nz)
        r = oracle.decompress_ex("deflate", s, len(want))
        assert (r[0], r[3]) == (0, want), nz
        assert zlib.decompress(s, -15) == want
    assert oracle.decompress_ex("deflate", streams.too_many_codeword_lengths(),
                                1000)[0] == 1
    # must be BAD_DATA, not INSUFFICIENT_SPACE (test_overread.c:70-91)
    assert oracle.decompress_ex("deflate", streams.overread_stream(), 128)[0] == 1


def test_trailing_bytes_semantics(oracle):
    """programs/test_trailing_bytes.c:74-142"""
    data = streams.trailing_bytes_input()
    for fmt in ("deflate", "zlib", "gzip"):
        comp = streams._zcompress(fmt, 6, data)
        for extra in (b"", b"\x01\x02\x03\x04"):
            r, ain, aout, out = oracle.decompress_ex(fmt, comp + extra, len(data))
            assert (r, ain, aout, out) == (0, len(comp), len(data), data)
            r, ain, aout, out = oracle.decompress_ex(fmt, comp + extra,
                                                     len(data), False)
            assert (r, ain, out) == (0, len(comp), data)
            # exact-fill rule
            assert oracle.decompress_ex(fmt, comp, len(data) + 1, False)[0] == 2
            assert oracle.decompress_ex(fmt, comp, len(data) - 1)[0] == 3


def test_golden_fixtures(oracle):
    """Streams produced by the real reference (oracle/make_golden.py)."""
    with open(os.path.join(GOLDEN, "golden.json")) as f:
        g = json.load(f)
    assert len(g["cases"]) >= 100
    for cs in g["cases"]:
        comp = bytes.fromhex(cs["stream"])
        r, ain, aout, out = oracle.decompress_ex(cs["fmt"], comp, cs["avail"],
                                                 cs["want_out"])
        assert r == cs["result"], cs["tag"]
        if r == 0:
            assert ain == cs["actual_in"], cs["tag"]
            if cs["want_out"]:
                assert aout == cs["actual_out"], cs["tag"]
            assert oracle.crc32(out) == cs["out_crc32"], cs["tag"]
    for cs in g["checksums"]:
        d = datagen.chunk(cs["idx"], cs["n"], cs["seed"])
        assert oracle.crc32(d, cs["crc_init"]) == cs["crc32"]
        assert oracle.adler32(d, cs["adler_init"]) == cs["adler32"]


def test_golden_64k_fixtures(oracle):
    """Config-sized streams (64 KiB chunks of the benchmark mix) compressed
    by the real reference, valid and damaged."""
    n = 0
    for fmt, s, avail, want, tag, exp in streams.golden_64k_cases(
            os.path.join(GOLDEN, "golden_64k.json")):
        r, ain, aout, out = oracle.decompress_ex(fmt, s, avail, want)
        assert r == exp["result"], tag
        if r == 0:
            assert ain == exp["actual_in"], tag
            if want:
                assert aout == exp["actual_out"], tag
            assert oracle.crc32(out) == exp["out_crc32"], tag
        n += 1
    assert n == 50


def test_slow_decompression_vectors(oracle):
    """programs/test_slow_decompression.c:18-108,128-129: streams of nothing
    but empty blocks must end in BAD_DATA or INSUFFICIENT_SPACE."""
    for s in (streams.empty_static_blocks(), streams.empty_dynamic_blocks()):
        assert len(s) == 4096
        for want in (True, False):
            assert oracle.decompress_ex("deflate", s, 10000, want)[0] in (1, 3)
    # one complete empty dynamic block, made final, is a valid stream of 0 bytes
    one = bytearray(streams.empty_dynamic_blocks(64))
    one[0] |= 1
    assert oracle.decompress_ex("deflate", bytes(one), 0)[:3] == (0, 12, 0)


def test_live_against_reference(oracle, ref):
    comp = lambda fmt, lvl, d: ref.compress(fmt, lvl, d)
    cases = streams.random_cases(11, 250, compress=comp)
    cases += streams.garbage_cases(12, 400)
    for fmt, s, avail, want, tag in cases:
        a = ref.decompress_ex(fmt, s, avail, want)
        b = oracle.decompress_ex(fmt, s, avail, want)
        assert a[0] == b[0], (tag, fmt, a[:3], b[:3])
        if a[0] == 0:
            assert a == b, tag
    for n in (0, 1, 4999, 5000, 5001, 65536, 1 << 20):
        for fmt in ("deflate", "zlib", "gzip"):
            assert oracle.bound(fmt, n) == ref.bound(fmt, n)


def test_oracle_compressor_roundtrip_and_bound(oracle):
    """The restated compressor policy: valid streams (checked by the restated
    decoder AND by zlib), within compress_bound, all levels / formats / the
    edge sizes of SURVEY.md §8(d)."""
    wb = {"deflate": -15, "zlib": 15, "gzip": 31}
    sizes = [0, 1, 18, 19, 20, 31, 32, 51, 52, 511, 512, 4999, 5000, 5001,
             65535, 65536, 65537, 125500]
    for i, n in enumerate(sizes):
        d = datagen.chunk(i, n, 0x0E110030) if n != 125500 else streams.litrunlen_input()
        for lvl in (0, 1, 3, 6, 9, 12):
            for fmt in ("deflate", "zlib", "gzip"):
                z = oracle.compress(fmt, lvl, d)
                assert z is not None and len(z) <= oracle.bound(fmt, n)
                assert zlib.decompress(z, wb[fmt]) == d
                r = oracle.decompress_ex(fmt, z, n)
                assert (r[0], r[1], r[3]) == (0, len(z), d)
                # "0 when it does not fit" (libdeflate.h:73-74)
                assert oracle.compress(fmt, lvl, d, len(z) - 1) is None


def test_oracle_compressor_tracks_reference(oracle, ref):
    """Sizes of the restatement vs the real reference on the 64 KiB mix:
    levels 2-9 follow the same parser / block-split / Huffman policy and land
    within 0.5 %; level 1 uses the hash-chain finder (the reference has a
    separate 2-way hash table there) and is allowed to be smaller."""
    chunks = [datagen.chunk(i, 65536, 0x0E110003) for i in range(8)]
    for lvl in (0, 1, 2, 4, 5, 6, 8, 9):
        ours = sum(len(oracle.compress("deflate", lvl, c)) for c in chunks)
        theirs = sum(len(ref.compress("deflate", lvl, c)) for c in chunks)
        for c in chunks[:3]:
            z = oracle.compress("gzip", lvl, c)
            assert ref.decompress_ex("gzip", z, len(c))[3] == c
        if lvl == 1:
            assert ours <= theirs * 1.02
        else:
            assert abs(ours - theirs) <= theirs * 0.005, (lvl, ours, theirs)


def test_config1_reference_benchmark(tmp_path):
    """BASELINE.json configs[0]: the unmodified programs/benchmark on a 1 MiB
    enwik-style buffer, level 6, reference CPU path only (plumbing check)."""
    import subprocess
    exe = os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "reftests",
                       "benchmark_ref")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/reftests/benchmark_ref not built")
    f = tmp_path / "enwik1m.txt"
    f.write_bytes(datagen.text_chunk(1 << 20, 0x0E110001))
    r = subprocess.run([exe, "-6", str(f)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    line = [l for l in r.stdout.splitlines() if "Compressed 1048576 =>" in l][0]
    pct = float(line.split("(")[1].split("%")[0])
    assert 25.0 < pct < 40.0, line      # target ratio 30-38 %


def test_stored_then_match_and_gzip_fields(oracle, ref):
    """The hand-built streams the GPU tests use: oracle == reference == zlib."""
    import zlib
    for s, want in streams.stored_then_match_streams():
        assert zlib.decompress(s, -15) == want
        for chk in (oracle, ref):
            assert chk.decompress_ex("deflate", s, len(want)) == (0, len(s), len(want), want)
    for s, want in streams.static_dynamic_static_streams():
        assert zlib.decompress(s, -15) == want
        for chk in (oracle, ref):
            assert chk.decompress_ex("deflate", s, len(want)) == (0, len(s), len(want), want)
            assert chk.decompress_ex("deflate", s[:-3], len(want))[0] != 0
    for s, want in streams.gzip_optional_field_streams():
        assert zlib.decompress(s, 31) == want
        for chk in (oracle, ref):
            assert chk.decompress_ex("gzip", s, len(want)) == (0, len(s), len(want), want)
        bad = s[:3] + bytes([s[3] | 0x20]) + s[4:]
        assert oracle.decompress_ex("gzip", bad, len(want))[0] == \
            ref.decompress_ex("gzip", bad, len(want))[0] == 1


def test_parallel_round_corner_streams(oracle, ref):
    """The long streams aimed at the GPU's parallel rounds: the oracle agrees
    with the reference (and zlib) on them, valid and invalid alike."""
    import zlib
    for name, s, want in streams.parallel_round_streams():
        assert zlib.decompress(s, -15) == want, name
        for chk in (oracle, ref):
            assert chk.decompress_ex("deflate", s, len(want)) == (0, len(s), len(want), want), name
        cut = s[:len(s) * 2 // 3]
        assert oracle.decompress_ex("deflate", cut, len(want))[0] == \
            ref.decompress_ex("deflate", cut, len(want))[0] != 0
    n_bad = 0
    for s in streams.bad_distance_streams():
        o = oracle.decompress_ex("deflate", s, 200000)
        r = ref.decompress_ex("deflate", s, 200000)
        assert o[:3] == r[:3]
        n_bad += o[0] == 1
    assert n_bad >= 2
