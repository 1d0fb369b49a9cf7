"""The reference's OWN unit-test programs (programs/test_*.c),
compiled from the reference sources where they lie and linked against
libdeflate_amd.so instead of libdeflate (oracle/Makefile `reftests`).  They
exercise the 21 drop-in symbols exactly the way a libdeflate user does:
host pointers, one call per buffer, zlib as their control implementation.
The binaries live in oracle/_ref/reftests/ (git-ignored, shipped by gpurun)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIR = os.path.join(ROOT, "oracle", "_ref", "reftests")
PROGRAMS = ["test_checksums", "test_custom_malloc", "test_incomplete_codes",
            "test_invalid_streams", "test_litrunlen_overflow", "test_overread",
            "test_trailing_bytes"]


@pytest.mark.parametrize("prog", PROGRAMS)
def test_reference_program(prog):
    exe = os.path.join(DIR, prog)
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/reftests not built (needs /root/reference at build time)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (prog, r.stdout[-2000:], r.stderr[-2000:])


def _enwik_file(tmp_path):
    from tests import datagen
    f = tmp_path / "enwik1m.txt"
    f.write_bytes(datagen.text_chunk(1 << 20, 0x0E110001))
    return str(f)


@pytest.mark.parametrize("args", [["-6"], ["-6", "-s", "65536"], ["-1", "-g"],
                                  ["-9", "-z", "-s", "4096"]])
def test_reference_benchmark_on_gpu_library(tmp_path, args):
    """programs/benchmark.c (unchanged) driving libdeflate_amd.so: it times
    and verifies the round trip of every chunk itself (benchmark.c:430-538)."""
    exe = os.path.join(DIR, "benchmark_amd")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/reftests not built")
    r = subprocess.run([exe] + args + [_enwik_file(tmp_path)], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "Compressed 1048576 =>" in r.stdout
    print(r.stdout[-300:])
