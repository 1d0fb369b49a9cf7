#!/usr/bin/env python3
"""fuzz_stream.py [seeds...] - randomized streams through the single-buffer
decompress calls with the many-wave path forced on for every size
(LDA_STREAM_PAR_MIN=0) and a random chunk size: contents of every kind cut at
odd places, levels of the reference (0-12) and of zlib, stored and static
blocks in between, valid / truncated / corrupted / short-output variants, a
stream stored inside a stream (every inner block header is a false candidate
for the block finder).  Result code, actual_in / actual_out and every byte
against the CPU oracle; exits non-zero on the first difference."""
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from tests import datagen, oracle_util, streams  # noqa: E402


def make_data(rng, n):
    out = bytearray()
    while len(out) < n:
        k = int(rng.integers(0, 10))
        ln = int(rng.integers(1, 200000))
        if k == 8:
            out += bytes(ln)
        elif k == 9:
            per = bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))
            out += (per * (ln // len(per) + 1))[:ln]
        else:
            out += datagen.chunk(k, ln, int(rng.integers(1 << 30)))
    return bytes(out[:n])


def make_stream(rng, ref, fmt, data):
    kind = int(rng.integers(0, 6))
    wb = {"deflate": -15, "zlib": 15, "gzip": 31}[fmt]
    if kind == 0 or ref is None:
        lvl = int(rng.choice([0, 1, 6, 9]))
        co = zlib.compressobj(lvl, zlib.DEFLATED, wb, int(rng.choice([1, 8, 9])),
                              int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED,
                                              zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE])))
        # flushes make empty stored blocks and byte-aligned block starts
        parts, pos = [], 0
        while pos < len(data):
            step = int(rng.integers(1, max(2, len(data))))
            parts.append(co.compress(data[pos:pos + step]))
            if rng.integers(0, 3) == 0:
                parts.append(co.flush(int(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH]))))
            pos += step
        parts.append(co.flush())
        return b"".join(parts), f"zlib{lvl}"
    lvl = int(rng.choice([0, 1, 2, 5, 6, 9, 10, 12]))
    return ref.compress(fmt, lvl, data), f"ref{lvl}"


def run(seeds, budget_s=None, log=print):
    import time
    from libdeflate_amd import api, binding
    oracle = oracle_util.load_oracle()
    ref = oracle_util.load_ref()
    os.environ["LDA_STREAM_PAR_MIN"] = "0"
    t0 = time.time()
    ncases = nbad = npar = 0
    dec = api.Decompressor()
    for k, seed in enumerate(seeds):
        if budget_s is not None and k and time.time() - t0 > budget_s:
            break
        rng = np.random.default_rng(seed)
        chunk = int(rng.choice([0, 4096, 8192, 32768]))
        if chunk:
            os.environ["LDA_STREAM_CHUNK"] = str(chunk)
        else:
            os.environ.pop("LDA_STREAM_CHUNK", None)
        binding.reload_env()
        bad = 0
        for it in range(6):
            n = int(rng.choice([0, 1, 100, 5000, 70000, 300000, 1 << 20, 3 << 20]))
            n = int(rng.integers(0, n + 1))
            data = make_data(rng, n)
            fmt = str(rng.choice(["deflate", "zlib", "gzip"]))
            z, tag = make_stream(rng, ref, fmt, data)
            if rng.integers(0, 4) == 0 and len(z) < (1 << 20):
                # the stream itself as the payload of stored blocks
                data, (z, tag) = z, (streams._zcompress(fmt, 0, z), tag + "/stored")
                n = len(data)
            variants = [("ok", z, n, True), ("exact", z, n, False), ("short", z, n + 3, False),
                        ("nospace", z, max(0, n - 1), True), ("tail", z + b"\x00\xff\x07", n, True)]
            if z:
                cut = int(rng.integers(0, len(z)))
                variants.append((f"trunc{cut}", z[:cut], n, True))
                b = bytearray(z)
                p = int(rng.integers(0, len(b)))
                b[p] ^= 1 << int(rng.integers(0, 8))
                variants.append((f"flip{p}", bytes(b), n, True))
            for name, s, avail, want in variants:
                g = dec.decompress_ex(fmt, s, avail, want)
                npar += binding.stream_stats()["parallel"]
                o = oracle.decompress_ex(fmt, s, avail, want)
                ok = g[0] == o[0]
                if ok and o[0] == 0:
                    ok = g[1] == o[1] and (not want or g[2] == o[2]) and g[3] == o[3]
                ncases += 1
                if not ok:
                    bad += 1
                    log("MISMATCH", seed, it, tag, fmt, name, n, len(s), g[:3], o[:3],
                        binding.stream_stats())
        log(f"seed={seed} chunk={chunk}: {bad} mismatches", flush=True)
        nbad += bad
        if bad:
            break
    dec.close()
    os.environ.pop("LDA_STREAM_PAR_MIN", None)
    os.environ.pop("LDA_STREAM_CHUNK", None)
    binding.reload_env()
    return ncases, nbad, npar


def main():
    seeds = [int(a) for a in sys.argv[1:]] or list(range(300, 310))
    n, bad, npar = run(seeds)
    print(f"{n} cases, {bad} mismatches, {npar} answered by the many-wave path")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
