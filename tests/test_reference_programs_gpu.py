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
