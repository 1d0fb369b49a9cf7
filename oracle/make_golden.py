"""Generates tests/golden/golden.json from the REAL reference
(oracle/_ref/libdeflate_ref.so, built by oracle/Makefile from /root/reference).
Run here (where /root/reference exists); the JSON is committed so the oracle
and the GPU path can be pinned on boxes that do not have the reference.

    python oracle/make_golden.py
"""
import base64
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests import datagen, oracle_util, streams  # noqa: E402


def orig_repro():
    """The `orig_repro` byte array of programs/test_slow_decompression.c."""
    src = open("/root/reference/programs/test_slow_decompression.c").read()
    m = re.search(r"orig_repro\[(\d+)\]\s*=(.*?);", src, re.S)
    data = bytes(int(t, 16) for t in re.findall(r"\\x([0-9a-fA-F]{2})", m.group(2)))
    assert len(data) == int(m.group(1))
    return data


def main():
    ref = oracle_util.load_ref()
    assert ref is not None, "build oracle/_ref first (make -C oracle)"
    comp = lambda fmt, lvl, d: ref.compress(fmt, lvl, d)
    cases = streams.random_cases(101, 60, compress=comp,
                                 sizes=[0, 1, 31, 32, 100, 1000, 5000, 9000])
    cases += streams.garbage_cases(102, 120)
    hand = [streams.incomplete_empty_offset_code()[0],
            streams.incomplete_singleton_litlen()[0],
            streams.incomplete_singleton_offset(False)[0],
            streams.incomplete_singleton_offset(True)[0],
            streams.too_many_codeword_lengths(), streams.overread_stream()]
    for i, s in enumerate(hand):
        cases.append(("deflate", s, 128, True, f"hand{i}"))
    # test_slow_decompression.c: the two generators (4096-byte input, 10000
    # bytes of output space, :426-453) and the blob of issue #33 (:176-424),
    # which is data of the reference's test and is read from there
    slow = [("slow_static", streams.empty_static_blocks(4096)),
            ("slow_dynamic", streams.empty_dynamic_blocks(4096)),
            ("slow_orig_repro", orig_repro())]
    for tag, s in slow:
        for want in (True, False):
            cases.append(("deflate", s, 10000, want, tag))
    # SHORT_OUTPUT (exact fill asked for, the stream is shorter) and
    # INSUFFICIENT_SPACE one byte below, per format and producer level
    for i, (n, lvl) in enumerate([(1, 1), (100, 6), (5000, 9), (9000, 12), (300, 0)]):
        d = datagen.chunk(40 + i, n, 0x0E110020)
        for fmt in ("deflate", "gzip") if i % 2 else ("zlib", "deflate"):
            z = ref.compress(fmt, lvl, d)
            cases.append((fmt, z, n + 1 + 7 * i, False, f"short{i}/{fmt}"))
            cases.append((fmt, z, n - 1, False, f"nospace{i}/{fmt}"))
    out = []
    for fmt, s, avail, want, tag in cases:
        r, ain, aout, data = ref.decompress_ex(fmt, s, avail, want)
        out.append({"fmt": fmt, "stream": s.hex(), "avail": avail,
                    "want_out": want, "tag": tag, "result": r,
                    "actual_in": ain if r == 0 else 0,
                    "actual_out": aout if r == 0 else 0,
                    "out_crc32": ref.crc32(data) if r == 0 else 0})
    sums = []
    for idx, n in enumerate([0, 1, 15, 16, 17, 1023, 1024, 1025, 5552, 5553,
                             65535, 65536, 65537]):
        d = datagen.chunk(idx, n, 0x0E110000)
        ci, ai = (idx * 2654435761) & 0xFFFFFFFF, ((idx * 40503) % 65521) << 16 | (idx * 9973) % 65521
        sums.append({"idx": idx, "n": n, "seed": 0x0E110000, "crc_init": ci,
                     "adler_init": ai, "crc32": ref.crc32(d, ci),
                     "adler32": ref.adler32(d, ai)})
    # config-sized streams: 64 KiB chunks of the benchmark mix compressed by
    # the reference (levels 1/6/9/12, the three formats), valid and damaged;
    # kept in their own file (base64) so the small cases stay readable
    big = []
    for i in range(10):
        d = datagen.chunk(i, 65536, 0x0E110004)
        fmt = ("gzip", "zlib", "deflate")[i % 3]
        lvl = (6, 1, 9, 12, 6)[i % 5]
        z = ref.compress(fmt, lvl, d)
        # (name, keep the first `cut` bytes, flip bit 4 of byte `flip`, avail, want)
        variants = [("ok", None, None, 65536, True), ("exact", None, None, 65536, False),
                    ("short", None, None, 65537, False), ("nospace", None, None, 65535, True)]
        variants.append(("trunc", len(z) * 2 // 3, None, 65536, True) if i % 2 else
                        ("flip", None, len(z) // 2, 65536, True))
        vs = []
        for name, cut, flip, avail, want in variants:
            s = streams.damage(z, cut, flip)
            r, ain, aout, data = ref.decompress_ex(fmt, s, avail, want)
            vs.append({"name": name, "cut": cut, "flip": flip, "avail": avail,
                       "want_out": want, "result": r,
                       "actual_in": ain if r == 0 else 0,
                       "actual_out": aout if r == 0 else 0,
                       "out_crc32": ref.crc32(data) if r == 0 else 0})
        big.append({"fmt": fmt, "level": lvl, "tag": f"big{i}/l{lvl}",
                    "stream_b64": base64.b64encode(z).decode(), "variants": vs})
    with open(os.path.join(ROOT, "tests", "golden", "golden_64k.json"), "w") as f:
        json.dump({"generator": "oracle/make_golden.py",
                   "reference": "libdeflate v1.25 (/root/reference), gcc -O2",
                   "streams": big}, f, indent=0)
    print("wrote golden_64k.json", len(big), "streams")
    path = os.path.join(ROOT, "tests", "golden", "golden.json")
    with open(path, "w") as f:
        json.dump({"generator": "oracle/make_golden.py",
                   "reference": "libdeflate v1.25 (/root/reference), gcc -O2",
                   "cases": out, "checksums": sums}, f, indent=0)
    print("wrote", path, len(out), "cases")


if __name__ == "__main__":
    main()
