"""CPU-side checks of the boundary: the shared library loads and exports every
symbol include/libdeflate_amd.h declares, the header matches the binding, and
the calls that need no device behave (bounds, NULL rules).  No GPU compute."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    from libdeflate_amd import binding
    if not os.path.exists(binding.LIB_PATH):
        g.build()
    return binding.load()


def test_header_symbols_exported(lib):
    from libdeflate_amd import binding
    hdr = open(os.path.join(ROOT, "include", "libdeflate_amd.h")).read()
    declared = set(re.findall(r"^(libdeflate_[a-z0-9_]+)\(", hdr, re.M))
    assert len([s for s in declared if not s.startswith("libdeflate_amd_")]) == 21
    assert declared == set(binding.DROPIN_SYMBOLS) | set(binding.BATCH_SYMBOLS)
    assert binding.MISSING == []
    out = subprocess.run(["nm", "-D", "--defined-only", binding.LIB_PATH],
                         capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (libdeflate_\w+)", out))
    assert declared <= exported
    # nothing of the oracle leaks into the product
    assert "oracle" not in out


def test_bounds_and_null_rules_without_device(lib, oracle):
    # pure functions of n: lib/deflate_compress.c:4087-4135
    for n in (0, 1, 4999, 5000, 5001, 65536, 1 << 20, (1 << 32) + 5):
        for f in ("deflate", "zlib", "gzip"):
            got = getattr(lib, f"libdeflate_{f}_compress_bound")(None, n)
            assert got == oracle.bound(f, n)
    assert lib.libdeflate_deflate_compress_bound(None, 65536) == 65606
    assert lib.libdeflate_gzip_compress_bound(None, 4096) == 4119
    # NULL buffer -> initial value, before any device work
    assert lib.libdeflate_crc32(123, None, 99) == 0
    assert lib.libdeflate_adler32(123, None, 99) == 1
    lib.libdeflate_free_compressor(None)
    lib.libdeflate_free_decompressor(None)


def test_fails_loudly_without_gpu(lib):
    """No CPU fallback: on a box without a gfx950 device the allocators return
    NULL and report why."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert lib.libdeflate_amd_device_ready() != 0
    assert not lib.libdeflate_alloc_compressor(6)
    assert not lib.libdeflate_alloc_decompressor()
    assert lib.libdeflate_amd_last_error()


def test_alloc_argument_validation(lib):
    # level outside [-1, 12] -> NULL (libdeflate.h:49-50), checked before the
    # device is touched; bad sizeof_options -> NULL (deflate_compress.c:3885)
    assert not lib.libdeflate_alloc_compressor(13)
    assert not lib.libdeflate_alloc_compressor(-2)

    class Opt(ctypes.Structure):
        _fields_ = [("sizeof_options", ctypes.c_size_t),
                    ("malloc_func", ctypes.c_void_p),
                    ("free_func", ctypes.c_void_p)]
    bad = Opt(8, None, None)
    assert not lib.libdeflate_alloc_compressor_ex(6, ctypes.byref(bad))
    assert not lib.libdeflate_alloc_decompressor_ex(ctypes.byref(bad))


def test_product_does_not_reference_oracle():
    """The product path must never import/link the checker."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "libdeflate_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.lower() or f == "Makefile", (dirpath, f)


def test_compress_kernels_keep_their_register_budget():
    """The compress kernels are issue bound and were held at 128 VGPRs (8 spilled) by ~55
    loop-invariant constants that MachineLICM hoists out of the per-buffer loop; the
    Makefile builds the two compress objects without that pass.  A
    toolchain or a source change that brings the pressure back should say so here, not
    in a slower bench: the report of the compile itself, no GPU needed."""
    csrc = os.path.join(os.path.dirname(__file__), "..", "libdeflate_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    assert re.search(r"^NOLICM \?= .*\bdeflate_kernel\b.*\bdeflate_small\b", mk, re.M)
    assert re.search(r"\$\(NOLICM\).*: CXXFLAGS \+= -mllvm -disable-machine-licm", mk)
    flags = ["-mllvm", "-disable-machine-licm"]
    if "deflate_kernel" in re.search(r"^MAXILP \?= (.*)$", mk, re.M).group(1).split():
        flags += ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]    # (the Makefile's flags for this object)
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                        "-fvisibility=hidden", "-ffp-contract=off", *flags,
                        "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c",
                        "deflate_kernel.hip", "-o", os.devnull],
                       cwd=csrc, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = r.stderr
    kernels = re.findall(r"Function Name: (lda_deflate_\w+)", rep)
    assert set(kernels) >= {"lda_deflate_batch_kernel", "lda_deflate_opt_kernel"}
    vgprs = [int(x) for x in re.findall(r" VGPRs: (\d+)", rep)]
    spills = [int(x) for x in re.findall(r"VGPRs Spill: (\d+)", rep)]
    assert vgprs and max(vgprs) <= 104, vgprs     # round 5: 85 / 83; round 6 (max-ilp scheduling): 99 / 97
    assert spills and max(spills) == 0, spills


def _device_asm(src, *flags):
    csrc = os.path.join(os.path.dirname(__file__), "..", "libdeflate_amd", "csrc")
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                        *flags, "--cuda-device-only", "-S", src, "-o", "-"],
                       cwd=csrc, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def _kernel_body(asm, name):
    m = re.search(r"^%s:[^\n]*\n(.*?)\n\s*s_endpgm" % re.escape(name), asm, re.M | re.S)
    assert m, name
    return m.group(1)


def test_visibility_probe_uses_the_products_access_pattern():
    """the self-check that admits a device must test the access the product
    makes.  The inflate kernel reads back output bytes another lane of the wave stored
    with PLAIN global loads (served by the L1) behind an explicit `s_waitcnt vmcnt(0)`
    (global_stores_visible()); a volatile access would compile to `sc0 sc1`, system
    scope, and test something else.  Checked on the ISA of both, no GPU needed: the
    probe's and the product's byte loads / stores carry no sc / nt bits, and both
    contain the wait as inline asm."""
    probe = _kernel_body(_device_asm("selfcheck_kernels.hip"), "lda_selfcheck_visibility_kernel")
    mem = [l.strip() for l in probe.splitlines()
           if re.match(r"\s*(global|flat|buffer)_(load|store)", l)]
    assert len(mem) >= 4, mem
    assert not [l for l in mem if re.search(r"\b(sc0|sc1|nt)\b", l)], mem
    assert all(l.startswith("global_") for l in mem), mem
    assert re.search(r";;#ASMSTART\s*\n\s*s_waitcnt vmcnt\(0\)", probe)
    from libdeflate_amd import binding  # noqa: F401  (the Makefile's flags for this object)
    mk = open(os.path.join(os.path.dirname(__file__), "..", "libdeflate_amd", "csrc", "Makefile")).read()
    nolicm = re.search(r"^NOLICM \?= (.*)$", mk, re.M).group(1).split()
    flags = ("-mllvm", "-disable-machine-licm") if "inflate_kernel" in nolicm else ()
    wave = _kernel_body(_device_asm("inflate_kernel.hip", *flags), "lda_inflate_wave_kernel")
    far = [l.strip() for l in wave.splitlines() if re.match(r"\s*global_load_ubyte", l)]
    assert far, "no byte loads from the output in the wave kernel?"
    assert not [l for l in far if re.search(r"\b(sc0|sc1|nt)\b", l)], far
    # the round start, a batch of slots with a just-stored source, the block end, a stored block
    assert len(re.findall(r";;#ASMSTART\s*\n\s*s_waitcnt vmcnt\(0\)", wave)) >= 4
