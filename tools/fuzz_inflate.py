#!/usr/bin/env python3
"""fuzz_inflate.py [seeds...] - randomized valid / truncated / corrupted streams
of sizes up to 300 KB through the GPU decompressor (both mappings), compared
with the CPU oracle: result code, actual_in / actual_out and every byte.  A
longer-running companion of tests/test_inflate_gpu.py (same case generator);
prints a summary line per seed and exits non-zero on the first difference."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from tests import oracle_util, streams  # noqa: E402


def run(seeds, budget_s=None, log=print):
    """-> (cases run, mismatches).  Stops early (after at least one seed per
    mapping) once `budget_s` seconds have passed."""
    import time
    from libdeflate_amd import api, binding
    oracle = oracle_util.load_oracle()
    ref = oracle_util.load_ref()
    sizes = [0, 1, 31, 100, 1000, 4096, 5000, 20000, 65536, 70000, 150000, 300000]
    t0 = time.time()
    ncases = nbad = 0
    for mode in ("1", "0"):
        os.environ["LDA_INFLATE_PAR"] = mode
        binding.reload_env()
        dec = api.Decompressor()
        for k, seed in enumerate(seeds):
            if budget_s is not None and k and time.time() - t0 > budget_s * (1 if mode == "0" else 0.5):
                break
            # odd seeds: streams made by the real reference (levels map to its
            # 0-12 range: its block splitting and min-cost parses differ from
            # zlib's); even seeds: Python's zlib
            comp = None
            if ref is not None and seed % 2:
                comp = lambda fmt, lvl, d: ref.compress(fmt, {0: 0, 1: 2, 3: 5, 6: 8, 9: 12}[lvl], d)
            cases = streams.random_cases(seed, 120, compress=comp, sizes=sizes)
            groups = {}
            for cs in cases:
                groups.setdefault((cs[0], cs[3]), []).append(cs)
            bad = 0
            for (fmt, want), group in groups.items():
                got = dec.decompress_batch_host(fmt, [g[1] for g in group],
                                                [g[2] for g in group], want)
                for cs, g in zip(group, got):
                    o = oracle.decompress_ex(fmt, cs[1], cs[2], want)
                    ok = g[0] == o[0]
                    if ok and o[0] == 0:
                        ok = g[1] == o[1] and (not want or g[2] == o[2]) and g[3] == o[3]
                    if not ok:
                        bad += 1
                        log("MISMATCH", mode, seed, cs[4], fmt, want, g[:3], o[:3])
            log(f"par={mode} seed={seed}: {len(cases)} cases, {bad} mismatches", flush=True)
            ncases += len(cases)
            nbad += bad
            if bad:
                break
        dec.close()
    os.environ.pop("LDA_INFLATE_PAR", None)
    binding.reload_env()
    return ncases, nbad


def main():
    seeds = [int(a) for a in sys.argv[1:]] or [101, 102, 103]
    _, bad = run(seeds)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
