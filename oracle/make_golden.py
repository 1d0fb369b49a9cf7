"""Generates tests/golden/golden.json from the REAL reference
(oracle/_ref/libdeflate_ref.so, built by oracle/Makefile from /root/reference).
Run here (where /root/reference exists); the JSON is committed so the oracle
and the GPU path can be pinned on boxes that do not have the reference.

    python oracle/make_golden.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests import datagen, oracle_util, streams  # noqa: E402


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
    path = os.path.join(ROOT, "tests", "golden", "golden.json")
    with open(path, "w") as f:
        json.dump({"generator": "oracle/make_golden.py",
                   "reference": "libdeflate v1.25 (/root/reference), gcc -O2",
                   "cases": out, "checksums": sums}, f, indent=0)
    print("wrote", path, len(out), "cases")


if __name__ == "__main__":
    main()
