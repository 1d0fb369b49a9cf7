"""The two fuzz drivers of tools/ inside the GPU suite, on a time budget: the
decompressor against the oracle (random valid / truncated / corrupted streams
up to 300 KB, both stream mappings: result code, actual_in / actual_out, every
byte) and the compressor (every level and format, sizes around the tile /
block / segment limits, both kernels and the segmented single-buffer path:
every stream decoded with zlib and held against compress_bound).  The seeds
change from run to run of the suite only if FUZZ_SEED is set; a longer run is
`python tools/fuzz_inflate.py 1 2 3 ...`."""
import os

import pytest

pytestmark = pytest.mark.gpu

BUDGET_S = float(os.environ.get("FUZZ_BUDGET_S", "20"))
BASE = int(os.environ.get("FUZZ_SEED", "7000"))


def test_fuzz_inflate_against_oracle():
    from tools import fuzz_inflate
    n, bad = fuzz_inflate.run(list(range(BASE, BASE + 40)), BUDGET_S, log=lambda *a, **k: None)
    assert n >= 240 and bad == 0, (n, bad)


def test_fuzz_deflate_roundtrip():
    from tools import fuzz_deflate
    n, bad = fuzz_deflate.run(list(range(BASE, BASE + 200)), BUDGET_S, log=lambda *a, **k: None)
    assert n >= 24 and bad == 0, (n, bad)


def test_fuzz_stream_path_against_oracle():
    """tools/fuzz_stream.py: every single-buffer decompress through the
    many-wave path (forced on for all sizes, random chunk sizes): streams of
    every producer and block kind, damaged variants, streams stored inside
    streams; result codes and bytes against the oracle."""
    from tools import fuzz_stream
    msgs = []
    n, bad, npar = fuzz_stream.run(list(range(BASE + 500, BASE + 540)), BUDGET_S,
                                   log=lambda *a, **k: msgs.append(a))
    assert bad == 0, msgs[-5:]
    assert n >= 40 and npar >= n // 6, (n, npar)
