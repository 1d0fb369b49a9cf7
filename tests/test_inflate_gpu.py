"""GPU inflate (HIP) vs the oracle, through the C-ABI: result codes,
actual_in/actual_out and every output byte.  Mirrors the reference's
test_incomplete_codes / test_invalid_streams / test_overread /
test_trailing_bytes programs plus randomized and corrupted streams."""
import json
import os
import zlib

import numpy as np
import pytest

from libdeflate_amd import binding

from tests import datagen, streams

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def dec():
    from libdeflate_amd import api
    d = api.Decompressor()
    yield d
    d.close()


def _run_cases(dec, oracle, cases):
    """Group by (fmt, want_actual_out) and push each group as one batch."""
    groups = {}
    for cs in cases:
        groups.setdefault((cs[0], cs[3]), []).append(cs)
    for (fmt, want), group in groups.items():
        got = dec.decompress_batch_host(fmt, [g[1] for g in group],
                                        [g[2] for g in group], want)
        for cs, g in zip(group, got):
            o = oracle.decompress_ex(fmt, cs[1], cs[2], want)
            assert g[0] == o[0], (cs[4], fmt, "gpu", g[:3], "oracle", o[:3])
            if o[0] == 0:
                assert g[1] == o[1], (cs[4], "actual_in")
                if want:
                    assert g[2] == o[2], (cs[4], "actual_out")
                assert g[3] == o[3], (cs[4], "bytes differ")


def _device_batch_timed(dec, fmt, chunks, avail):
    """The chunks as one device batch (HBM to HBM): result codes and the
    milliseconds of the second of two runs (HIP events around the call)."""
    import torch
    dev = torch.device("cuda:0")
    n = len(chunks)
    offs, blob = [], bytearray()
    for c in chunks:
        offs.append(len(blob))
        blob += c
        blob += bytes(-len(blob) % 16)
    data = torch.frombuffer(bytearray(blob) + bytearray(64), dtype=torch.uint8).to(dev)
    in_off = torch.tensor(offs, dtype=torch.int64, device=dev)
    in_n = torch.tensor([len(c) for c in chunks], dtype=torch.int64, device=dev)
    slot = (avail + 15) // 16 * 16
    out = torch.zeros(n * slot + 64, dtype=torch.uint8, device=dev)
    out_off = torch.arange(n, dtype=torch.int64, device=dev) * slot
    out_av = torch.full((n,), avail, dtype=torch.int64, device=dev)
    res = torch.full((n,), -1, dtype=torch.int32, device=dev)
    ain = torch.zeros(n, dtype=torch.int64, device=dev)
    aout = torch.zeros(n, dtype=torch.int64, device=dev)
    ms = 0.0
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dec.decompress_batch(fmt, data, in_off, in_n, out, out_off, out_av, res, ain, aout,
                             stream=torch.cuda.current_stream())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
    return res.cpu().tolist(), ms


def test_reference_unit_test_vectors(dec, oracle):
    cases = []
    for i, (s, want) in enumerate([streams.incomplete_empty_offset_code(),
                                   streams.incomplete_singleton_litlen(),
                                   streams.incomplete_singleton_offset(False),
                                   streams.incomplete_singleton_offset(True)]):
        cases.append(("deflate", s, len(want), True, f"incomplete{i}"))
        cases.append(("deflate", s, len(want), False, f"incomplete{i}x"))
    cases.append(("deflate", streams.too_many_codeword_lengths(), 1000, True, "toomany"))
    cases.append(("deflate", streams.overread_stream(), 128, True, "overread"))
    _run_cases(dec, oracle, cases)
    # and the literal expectations of the reference tests
    r = dec.decompress_ex("deflate", streams.overread_stream(), 128)
    assert r[0] == 1
    s, want = streams.incomplete_singleton_offset(True)
    assert dec.decompress_ex("deflate", s, len(want))[3] == want


def test_trailing_bytes(dec, oracle):
    data = streams.trailing_bytes_input()
    for fmt in ("deflate", "zlib", "gzip"):
        comp = streams._zcompress(fmt, 6, data)
        for extra in (b"", b"\x01\x02\x03\x04"):
            assert dec.decompress_ex(fmt, comp + extra, len(data)) == \
                (0, len(comp), len(data), data)
            r, aout, out = dec.decompress(fmt, comp + extra, len(data), False)
            assert (r, out) == (0, data)
        assert dec.decompress(fmt, comp, len(data) + 1, False)[0] == 2
        assert dec.decompress(fmt, comp, len(data) - 1)[0] == 3


def test_golden_fixtures(dec):
    """Results of the real reference, committed in tests/golden/."""
    with open(os.path.join(GOLDEN, "golden.json")) as f:
        g = json.load(f)
    groups = {}
    for cs in g["cases"]:
        groups.setdefault((cs["fmt"], cs["want_out"]), []).append(cs)
    for (fmt, want), group in groups.items():
        got = dec.decompress_batch_host(
            fmt, [bytes.fromhex(c["stream"]) for c in group],
            [c["avail"] for c in group], want)
        for cs, r in zip(group, got):
            assert r[0] == cs["result"], cs["tag"]
            if r[0] == 0:
                assert r[1] == cs["actual_in"], cs["tag"]
                if want:
                    assert r[2] == cs["actual_out"], cs["tag"]
                assert zlib.crc32(r[3]) == cs["out_crc32"], cs["tag"]


def test_golden_64k_fixtures(dec):
    """64 KiB chunks of the benchmark mix compressed by the real reference
    (committed, so config-sized parity does not rest on oracle/_ref being on
    the box), valid and damaged, both mappings."""
    cases = list(streams.golden_64k_cases(os.path.join(GOLDEN, "golden_64k.json")))
    assert len(cases) == 50
    for mode in ("1", "0"):
        os.environ["LDA_INFLATE_PAR"] = mode
        binding.reload_env()
        for want in (True, False):
            grp = [c for c in cases if c[3] == want]
            for fmt in ("deflate", "zlib", "gzip"):
                g2 = [c for c in grp if c[0] == fmt]
                got = dec.decompress_batch_host(fmt, [c[1] for c in g2],
                                                [c[2] for c in g2], want)
                for c, r in zip(g2, got):
                    exp = c[5]
                    assert r[0] == exp["result"], (mode, c[4])
                    if r[0] == 0:
                        assert r[1] == exp["actual_in"], c[4]
                        if want:
                            assert r[2] == exp["actual_out"], c[4]
                        assert zlib.crc32(r[3]) == exp["out_crc32"], c[4]


def test_slow_decompression_vectors(dec, oracle):
    """programs/test_slow_decompression.c: streams that open a new Huffman
    block every few bits (:18-108) and the blob of issue #33 (in golden.json)
    must come back BAD_DATA or INSUFFICIENT_SPACE (:128-129) - in both
    mappings, with the oracle's verdict - and a whole batch of them must not
    keep the device for long: the block headers are the serial part of this
    decoder (lane 0 of the stream's wave)."""
    import time
    import torch
    vec = [streams.empty_static_blocks(), streams.empty_dynamic_blocks()]
    cases = []
    for i, s in enumerate(vec):
        for want in (True, False):
            cases.append(("deflate", s, 10000, want, f"slow{i}"))
    for mode in ("1", "0"):
        os.environ["LDA_INFLATE_PAR"] = mode
        binding.reload_env()
        _run_cases(dec, oracle, cases)
        for s in vec:
            assert dec.decompress_ex("deflate", s, 10000)[0] in (1, 3)
    os.environ.pop("LDA_INFLATE_PAR")
    binding.reload_env()
    # 4096 such streams as one device batch (the shape of BASELINE configs[3]
    # with the worst input there is): time it
    for name, s in (("static", vec[0]), ("dynamic", vec[1])):
        n = 4096
        got, ms = _device_batch_timed(dec, "deflate", [s] * n, 10000)
        assert all(r in (1, 3) for r in got), name
        print(f"slow_decompression batch, {name}: {n} x 4 KiB in {ms:.2f} ms "
              f"({n * 4096 / ms / 1e3:.1f} MB/s of input)")
        assert ms < 2000, (name, ms)


def test_randomized_vs_oracle(dec, oracle):
    from tests import oracle_util
    ref = oracle_util.load_ref()
    comp = (lambda f, l, d: ref.compress(f, l, d)) if ref else None
    _run_cases(dec, oracle, streams.random_cases(21, 300, compress=comp))
    _run_cases(dec, oracle, streams.garbage_cases(22, 600))


def test_levels_and_sizes(dec, oracle):
    """Edge-size sweep of SURVEY.md §8(d) x formats x producer levels."""
    from tests import oracle_util
    ref = oracle_util.load_ref()
    sizes = [0, 1, 18, 19, 20, 31, 32, 51, 52, 511, 512, 4999, 5000, 5001,
             65535, 65536, 65537, 300000]
    cases = []
    for i, n in enumerate(sizes):
        d = datagen.chunk(i, n, 0x0E110010)
        for fmt in ("deflate", "zlib", "gzip"):
            for lvl in (0, 1, 6, 9, 12):
                s = ref.compress(fmt, lvl, d) if ref else \
                    streams._zcompress(fmt, min(lvl, 9), d)
                cases.append((fmt, s, n, True, f"n{n}/l{lvl}"))
                cases.append((fmt, s, n, False, f"n{n}/l{lvl}/exact"))
    _run_cases(dec, oracle, cases)
    lit = streams.litrunlen_input()     # test_litrunlen_overflow.c
    for lvl in (3, 6, 12):
        s = ref.compress("deflate", lvl, lit) if ref else \
            streams._zcompress("deflate", min(lvl, 9), lit)
        assert dec.decompress_ex("deflate", s, len(lit))[3] == lit


def test_device_batch_bit_exact(dec):
    """Config-4 shape at reduced count: gzip streams of 64 KiB chunks,
    decoded from HBM to HBM, every byte compared."""
    import torch
    from tests import oracle_util
    ref = oracle_util.load_ref()
    n, size = 512, 65536
    chunks = datagen.batch(n, size, 0x0E110004, distinct=32)
    comp = [ref.compress("gzip", 6, c) if ref else streams._zcompress("gzip", 6, c)
            for c in chunks[:32]]
    comp = [comp[i % 32] for i in range(n)]
    offs, blob = [], bytearray()
    for c in comp:
        offs.append(len(blob))
        blob += c
        blob += bytes((-len(blob)) % 16)
    blob += bytes(64)
    d_in = torch.frombuffer(blob, dtype=torch.uint8).cuda()
    in_off = torch.tensor(offs, dtype=torch.int64).cuda()
    in_n = torch.tensor([len(c) for c in comp], dtype=torch.int64).cuda()
    d_out = torch.zeros(n * size, dtype=torch.uint8, device="cuda")
    out_off = torch.arange(n, dtype=torch.int64, device="cuda") * size
    out_av = torch.full((n,), size, dtype=torch.int64, device="cuda")
    res = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    dec.decompress_batch("gzip", d_in, in_off, in_n, d_out, out_off, out_av, res)
    torch.cuda.synchronize()
    assert res.cpu().numpy().tolist() == [0] * n
    got = d_out.cpu().numpy().tobytes()
    assert got == b"".join(chunks)


def test_many_streams_all_formats(dec):
    """Config-4 shape at reduced count (16384 streams -> several streams per
    wave): reference-compressed gzip / zlib / raw chunks of the 64 KiB mix,
    decoded HBM to HBM with exact-fill semantics, every byte compared."""
    import torch
    from tests import oracle_util
    ref = oracle_util.load_ref()
    n, size, distinct = 16384, 65536, 32
    chunks = datagen.batch(distinct, size, 0x0E110004)
    want = None
    for fmt in ("gzip", "zlib", "deflate"):
        comp = [ref.compress(fmt, 6, c) if ref else streams._zcompress(fmt, 6, c)
                for c in chunks]
        offs, blob, lens = [], bytearray(), []
        for i in range(n):
            z = comp[i % distinct]
            offs.append(len(blob))
            lens.append(len(z))
            blob += z
            blob += bytes((-len(blob)) % 16)
        blob += bytes(64)
        d_in = torch.frombuffer(blob, dtype=torch.uint8).cuda()
        in_off = torch.tensor(offs, dtype=torch.int64).cuda()
        in_n = torch.tensor(lens, dtype=torch.int64).cuda()
        d_out = torch.zeros(n * size, dtype=torch.uint8, device="cuda")
        out_off = torch.arange(n, dtype=torch.int64, device="cuda") * size
        out_av = torch.full((n,), size, dtype=torch.int64, device="cuda")
        res = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        dec.decompress_batch(fmt, d_in, in_off, in_n, d_out, out_off, out_av, res)
        torch.cuda.synchronize()
        assert int((res != 0).sum()) == 0
        if want is None:
            want = torch.frombuffer(bytearray(b"".join(chunks)), dtype=torch.uint8).cuda()
        assert torch.equal(d_out.view(n // distinct, distinct * size),
                           want.expand(n // distinct, distinct * size))
        del d_in, d_out


def _big_stream_cases(ref, seed):
    """Streams long enough for the sub-block parallel rounds, in the shapes
    that exercise their hand-overs to the sequential decoder."""
    rng = np.random.default_rng(seed)
    cases = []
    datas = [("text", datagen.chunk(0, 65536, seed)),
             ("binary", datagen.chunk(5, 65536, seed)),
             ("lowent", datagen.chunk(6, 40000, seed)),
             ("zeros", bytes(70000)),
             ("runs", (b"ab" * 700 + bytes(3000) + b"xyz" * 900) * 8),
             ("far", datagen.chunk(1, 33000, seed) * 3),      # distances ~32 KiB
             ("multi", b"".join(datagen.chunk(k, 30000, seed + k) for k in (0, 7, 5, 6, 0)))]
    for name, d in datas:
        for lvl in (1, 6, 12):
            for fmt in ("deflate", "gzip"):
                s = ref.compress(fmt, lvl, d)
                n = len(d)
                cases.append((fmt, s, n, True, f"{name}/l{lvl}"))
                cases.append((fmt, s, n, False, f"{name}/l{lvl}/exact"))
                # output buffers that end inside / just before the data
                for short in (1, 7, 300, n // 2):
                    cases.append((fmt, s, n - short, True, f"{name}/l{lvl}/short{short}"))
                cases.append((fmt, s, n + 100, False, f"{name}/l{lvl}/long"))
                # truncated input and flipped bytes deep inside the stream
                for cut in (len(s) // 2, len(s) - 9, len(s) - 40):
                    cases.append((fmt, s[:cut], n, True, f"{name}/l{lvl}/cut{cut}"))
                for _ in range(6):
                    pos = int(rng.integers(20, len(s) - 10))
                    bad = bytearray(s)
                    bad[pos] ^= 1 << int(rng.integers(0, 8))
                    cases.append((fmt, bytes(bad), n, True, f"{name}/l{lvl}/flip{pos}"))
    return cases


def test_parallel_rounds_vs_oracle(dec, oracle):
    """Result codes, sizes and bytes of long streams - valid, truncated,
    corrupted, with short and long output buffers - against the oracle."""
    from tests import oracle_util
    ref = oracle_util.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built")
    _run_cases(dec, oracle, _big_stream_cases(ref, 0x0E110021))


def test_both_mappings_agree(dec, oracle, monkeypatch):
    """wave-per-stream (parallel rounds) and lane-per-stream give the same
    result codes and bytes on the same batch."""
    from tests import oracle_util
    ref = oracle_util.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built")
    cases = _big_stream_cases(ref, 0x0E110022)[::3]
    outs = []
    for mode in ("1", "0"):
        monkeypatch.setenv("LDA_INFLATE_PAR", mode)
        binding.reload_env()        # the switches are read once, at load
        got = []
        for want in (True, False):
            grp = [c for c in cases if c[3] == want]
            got += dec.decompress_batch_host("gzip", [c[1] for c in grp if c[0] == "gzip"],
                                             [c[2] for c in grp if c[0] == "gzip"], want)
        outs.append(got)
    for a, b in zip(*outs):
        assert a[0] == b[0]
        if a[0] == 0:
            assert a[1:] == b[1:]


def test_stored_block_then_short_match(dec, oracle, monkeypatch):
    """[Huffman][non-empty stored][Huffman opening with a short-distance
    match], in both mappings, raw and inside gzip/zlib (a stale register
    history would give wrong bytes / a false checksum failure)."""
    import struct
    cases = []
    for i, (s, want) in enumerate(streams.stored_then_match_streams()):
        cases.append(("deflate", s, len(want), True, f"stored{i}"))
        cases.append(("deflate", s, len(want), False, f"stored{i}/exact"))
        g = b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\xff" + s + \
            struct.pack("<II", zlib.crc32(want), len(want))
        cases.append(("gzip", g, len(want), True, f"stored{i}/gzip"))
        z = b"\x78\x9c" + s + struct.pack(">I", zlib.adler32(want))
        cases.append(("zlib", z, len(want), True, f"stored{i}/zlib"))
    for mode in ("1", "0"):
        monkeypatch.setenv("LDA_INFLATE_PAR", mode)
        binding.reload_env()        # the switches are read once, at load
        _run_cases(dec, oracle, cases)
        for i, (s, want) in enumerate(streams.stored_then_match_streams()):
            assert dec.decompress_ex("deflate", s, len(want))[3] == want, (mode, i)


def test_static_tables_after_a_dynamic_block(dec, oracle, monkeypatch):
    """static -> stored -> dynamic -> stored -> static: the static tables are
    kept across static blocks and rebuilt after a dynamic block took their
    memory (`static_loaded`), in both mappings."""
    cases = []
    for i, (s, want) in enumerate(streams.static_dynamic_static_streams()):
        cases.append(("deflate", s, len(want), True, f"sds{i}"))
        cases.append(("deflate", s, len(want), False, f"sds{i}/exact"))
        cases.append(("deflate", s[:-3], len(want), True, f"sds{i}/cut"))
    for mode in ("1", "0"):
        monkeypatch.setenv("LDA_INFLATE_PAR", mode)
        binding.reload_env()
        _run_cases(dec, oracle, cases)
        for i, (s, want) in enumerate(streams.static_dynamic_static_streams()):
            assert dec.decompress_ex("deflate", s, len(want))[3] == want, (mode, i)


def test_codes_of_one_codeword_length(dec, oracle, monkeypatch):
    """Huffman blocks whose codewords all have (nearly) one length - bytes
    drawn evenly from 2^k values, zlib's Huffman-only strategy: a parse started
    at a wrong bit never falls in step, the rounds' passes would fix one lane
    each (par_phase_starts() in csrc/inflate_kernel.hip parses every lane from
    each possible start once instead).  Bytes and codes as the oracle's, both
    mappings, valid / truncated / short output; and one large stream of them
    through the many-wave path."""
    rng = np.random.default_rng(0xC0DE)
    cases, wants = [], []
    for vals, n in ((128, 70000), (64, 30000), (16, 9000), (2, 5000), (200, 66000), (256, 40000)):
        data = rng.integers(0, vals, n, dtype=np.uint8).tobytes()
        # a match now and then: a token that overhangs the next lane's first bits
        data = data[:n // 2] + data[100:400] + data[n // 2:]
        co = zlib.compressobj(6, zlib.DEFLATED, -15, 9, zlib.Z_HUFFMAN_ONLY if vals != 200 else zlib.Z_DEFAULT_STRATEGY)
        z = co.compress(data) + co.flush()
        assert zlib.decompress(z, -15) == data
        wants.append((z, data))
        cases.append(("deflate", z, len(data), True, f"onelen{vals}"))
        cases.append(("deflate", z, len(data), False, f"onelen{vals}/exact"))
        cases.append(("deflate", z, len(data) - 1, True, f"onelen{vals}/short"))
        cases.append(("deflate", z[:len(z) * 3 // 4], len(data), True, f"onelen{vals}/cut"))
    for mode in ("1", "0"):
        monkeypatch.setenv("LDA_INFLATE_PAR", mode)
        binding.reload_env()
        _run_cases(dec, oracle, cases)
        for z, data in wants:
            assert dec.decompress_ex("deflate", z, len(data))[3] == data, mode
    big = rng.integers(0, 128, 3 << 20, dtype=np.uint8).tobytes()
    co = zlib.compressobj(6, zlib.DEFLATED, 31, 9, zlib.Z_HUFFMAN_ONLY)
    zb = co.compress(big) + co.flush()
    r = dec.decompress_ex("gzip", zb, len(big))
    assert r == (0, len(zb), len(big), big), binding.stream_stats()


def test_table_bits_follow_the_block_size(dec, oracle, monkeypatch):
    """A block's primary tables have 10 / 8 bits when it has 8 KiB of input
    ahead of it and the block before was not a small one, else 9 / 7
    (LIT_TB / TB_BIG_BLOCK in csrc/inflate_kernel.hip).  Streams whose blocks
    alternate between the two - cut by flushes - over bytes and distances with
    a long-tailed distribution, so that codewords of 10, 11 and more bits (and
    offset codewords of 8 and 9) occur under both table sizes; both mappings,
    valid / truncated / short output, against the oracle."""
    rng = np.random.default_rng(0x7AB1E)

    def skewed(n):
        # geometric byte values: code lengths from 1 to 15 bits
        v = np.minimum(rng.geometric(0.08, n) - 1, 255).astype(np.uint8)
        b = bytearray(v.tobytes())
        # matches at geometric distances (long offset codewords are the rare ones)
        for _ in range(n // 40):
            d = int(min(rng.geometric(0.0008), 30000, n - 40)) + 1
            at = int(rng.integers(d, n - 20))
            ln = int(rng.integers(3, 18))
            b[at:at + ln] = b[at - d:at - d + ln]
        return bytes(b)

    cases, wants = [], []
    for k, sizes in enumerate(((60000, 3000, 50000, 40000), (2000, 2000, 70000, 1000, 30000),
                               (30000,), (9000, 9000, 9000), (100000, 500, 500, 100000))):
        co = zlib.compressobj(6 if k % 2 else 9, zlib.DEFLATED, -15)
        z, data = b"", b""
        for j, n in enumerate(sizes):
            part = skewed(n)
            data += part
            z += co.compress(part) + co.flush(zlib.Z_FULL_FLUSH if j % 2 else zlib.Z_SYNC_FLUSH)
        z += co.flush()
        assert zlib.decompress(z, -15) == data
        wants.append((z, data))
        cases.append(("deflate", z, len(data), True, f"tb{k}"))
        cases.append(("deflate", z, len(data), False, f"tb{k}/exact"))
        cases.append(("deflate", z, len(data) - 3, True, f"tb{k}/short"))
        cases.append(("deflate", z[:len(z) * 2 // 3], len(data), True, f"tb{k}/cut"))
        for _ in range(4):
            bad = bytearray(z)
            bad[int(rng.integers(10, len(z) - 10))] ^= 1 << int(rng.integers(0, 8))
            cases.append(("deflate", bytes(bad), len(data), True, f"tb{k}/flip"))
    for mode in ("1", "0"):
        monkeypatch.setenv("LDA_INFLATE_PAR", mode)
        binding.reload_env()
        _run_cases(dec, oracle, cases)
        for z, data in wants:
            assert dec.decompress_ex("deflate", z, len(data))[3] == data, mode


def test_parallel_round_corner_streams(dec, oracle, monkeypatch):
    """Streams aimed at the wave-per-stream rounds (more tokens in a piece
    than a lane records, copies inside a 64-byte slot, sources older than the
    LDS mirror, a distance reaching back before the stream that only the copy
    phase sees), in both mappings and with short output buffers."""
    cases = []
    for name, s, want in streams.parallel_round_streams():
        cases.append(("deflate", s, len(want), True, name))
        cases.append(("deflate", s, len(want), False, name + "/exact"))
        cases.append(("deflate", s, len(want) + 100, True, name + "/room"))
        cases.append(("deflate", s, len(want) - 1, True, name + "/short"))
        cases.append(("deflate", s[:len(s) * 2 // 3], len(want), True, name + "/cut"))
    for i, s in enumerate(streams.bad_distance_streams()):
        cases.append(("deflate", s, 200000, True, f"baddist{i}"))
    for mode in ("1", "0"):
        monkeypatch.setenv("LDA_INFLATE_PAR", mode)
        binding.reload_env()        # the switches are read once, at load
        _run_cases(dec, oracle, cases)
        for name, s, want in streams.parallel_round_streams():
            assert dec.decompress_ex("deflate", s, len(want))[3] == want, (mode, name)


def test_gzip_optional_header_fields(dec, oracle):
    """Valid members with FEXTRA / FNAME / FCOMMENT / FHCRC are decoded, not
    just rejected consistently (lib/gzip_decompress.c:69-100)."""
    cases = []
    for i, (s, want) in enumerate(streams.gzip_optional_field_streams()):
        cases.append(("gzip", s, len(want), True, f"flg{2 * i}"))
        cases.append(("gzip", s + b"trailing", len(want), False, f"flg{2 * i}/exact"))
        # truncated inside the optional fields / with a reserved flag bit
        cases.append(("gzip", s[:14], len(want), True, f"flg{2 * i}/cut"))
        cases.append(("gzip", s[:3] + bytes([s[3] | 0x20]) + s[4:], len(want), True,
                      f"flg{2 * i}/reserved"))
    _run_cases(dec, oracle, cases)
    for s, want in streams.gzip_optional_field_streams():
        assert dec.decompress_ex("gzip", s, len(want)) == (0, len(s), len(want), want)


def test_config4_full_count(dec):
    """BASELINE configs[3] at its full count: 65 536 reference-compressed gzip
    streams of the 64 KiB mix (4 GiB out), exact-fill mode, every byte
    compared on the device; the streams beyond the resident grid are handed
    out dynamically."""
    import torch
    from tests import oracle_util
    ref = oracle_util.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built")
    n, size, distinct = 65536, 65536, 64
    chunks = datagen.batch(distinct, size, 0x0E110004)
    comp = [ref.compress("gzip", 6, c) for c in chunks]
    offs, blob = [], bytearray()
    for z in comp:
        offs.append(len(blob))
        blob += z
        blob += bytes((-len(blob)) % 16)
    blen = len(blob)
    reps = n // distinct
    d_in = torch.cat([torch.frombuffer(blob, dtype=torch.uint8).cuda().repeat(reps),
                      torch.zeros(64, dtype=torch.uint8, device="cuda")])
    gi = torch.arange(n, device="cuda")
    in_off = torch.tensor(offs, dtype=torch.int64).cuda()[gi % distinct] + (gi // distinct) * blen
    in_n = torch.tensor([len(z) for z in comp], dtype=torch.int64).cuda()[gi % distinct]
    d_out = torch.zeros(n * size, dtype=torch.uint8, device="cuda")
    out_off = torch.arange(n, dtype=torch.int64, device="cuda") * size
    out_av = torch.full((n,), size, dtype=torch.int64, device="cuda")
    res = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    dec.decompress_batch("gzip", d_in, in_off, in_n, d_out, out_off, out_av, res)
    torch.cuda.synchronize()
    assert int((res != 0).sum()) == 0
    want = torch.frombuffer(bytearray(b"".join(chunks)), dtype=torch.uint8).cuda()
    assert torch.equal(d_out.view(reps, distinct * size),
                       want.expand(reps, distinct * size))


def _bgzf_member(data, level=6):
    """One BGZF block: gzip member whose FEXTRA carries 'BC' = size - 1."""
    import struct
    body = streams._zcompress("deflate", level, data)
    bsize = 18 + len(body) + 8
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" +
            struct.pack("<H", bsize - 1) + body +
            struct.pack("<II", zlib.crc32(data), len(data)))


def test_multi_member_gzip(dec, ref):
    """Concatenated members, the way programs/gzip.c:236-299 handles them:
    the reference decoded member after member is the expectation; BGZF-style
    members (sized headers) go to the device as one batch, plain
    concatenations member by member."""
    def ref_loop(buf, avail):
        pos, out = 0, b""
        while pos < len(buf):
            r, ain, aout, o = ref.decompress_ex("gzip", buf[pos:], avail - len(out))
            if r != 0:
                return r, out
            pos += ain
            out += o
        return 0, out
    parts = [datagen.chunk(i, n, 0x0E110050) for i, n in
             enumerate([65280, 1, 0, 40000, 65280, 777, 65280, 12345] * 8)]
    plain = b"".join(parts)
    bgzf = b"".join(_bgzf_member(p) for p in parts)
    assert ref_loop(bgzf, len(plain)) == (0, plain)
    r, ain, aout, nm, out = dec.gzip_decompress_members(bgzf, len(plain))
    assert (r, ain, aout, nm) == (0, len(bgzf), len(plain), len(parts)) and out == plain
    # ordinary members (no size in the header): the sequential path
    cat = b"".join(streams._zcompress("gzip", 6, p) for p in parts[:9])
    want = b"".join(parts[:9])
    assert ref_loop(cat, len(want)) == (0, want)
    r, ain, aout, nm, out = dec.gzip_decompress_members(cat, len(want) + 10)
    assert (r, ain, aout, nm) == (0, len(cat), len(want), 9) and out == want
    # failures agree with the reference's loop: output too small, corrupt member,
    # trailing garbage after the last member
    assert dec.gzip_decompress_members(bgzf, len(plain) - 1)[0] == 3
    assert dec.gzip_decompress_members(cat, len(want) - 1)[0] == ref_loop(cat, len(want) - 1)[0] == 3
    bad = bytearray(bgzf)
    bad[len(bgzf) // 2] ^= 0x10
    assert dec.gzip_decompress_members(bytes(bad), len(plain))[0] != 0
    assert ref_loop(bytes(bad), len(plain))[0] != 0
    assert dec.gzip_decompress_members(cat + b"junk", len(want))[0] == \
        ref_loop(cat + b"junk", len(want))[0] == 1
    # an ordinary member behind an indexed (BGZF) prefix: the prefix is one
    # batch, the rest continues member after member from where the index ends
    mixed = b"".join(_bgzf_member(p) for p in parts[:5]) + \
        streams._zcompress("gzip", 6, parts[5]) + _bgzf_member(parts[6])
    wantm = b"".join(parts[:7])
    assert ref_loop(mixed, len(wantm)) == (0, wantm)
    r, ain, aout, nm, out = dec.gzip_decompress_members(mixed, len(wantm))
    assert (r, ain, aout, nm) == (0, len(mixed), len(wantm), 7) and out == wantm
    # a single ordinary member is the one-member case of the same call
    one = streams._zcompress("gzip", 6, parts[0])
    assert dec.gzip_decompress_members(one, len(parts[0]))[:4] == (0, len(one), len(parts[0]), 1)
