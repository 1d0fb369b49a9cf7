"""ctypes binding of libdeflate_amd.so (the C-ABI in include/libdeflate_amd.h).

This is the stub a Python caller of the reference would add (the reference's
own bindings list, README.md:146-157, are all thin FFI layers over
libdeflate.h).  It only declares signatures; all work happens in the HIP
library.  There is no fallback: if the shared object is missing the import
fails with instructions to build it.
"""
import ctypes
import os
from ctypes import (POINTER, c_char_p, c_int, c_int32, c_size_t, c_uint32,
                    c_uint64, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
# LIBDEFLATE_AMD_LIB selects another build of the same library (e.g. the
# phase-profiling build libdeflate_amd_prof.so); never a different backend.
LIB_PATH = os.environ.get("LIBDEFLATE_AMD_LIB",
                          os.path.join(_HERE, "libdeflate_amd.so"))

SUCCESS, BAD_DATA, SHORT_OUTPUT, INSUFFICIENT_SPACE = 0, 1, 2, 3
FMT_DEFLATE, FMT_ZLIB, FMT_GZIP = 0, 1, 2
FORMATS = {"deflate": FMT_DEFLATE, "zlib": FMT_ZLIB, "gzip": FMT_GZIP}

# every symbol include/libdeflate_amd.h declares
DROPIN_SYMBOLS = [
    "libdeflate_alloc_compressor", "libdeflate_alloc_compressor_ex",
    "libdeflate_deflate_compress", "libdeflate_deflate_compress_bound",
    "libdeflate_zlib_compress", "libdeflate_zlib_compress_bound",
    "libdeflate_gzip_compress", "libdeflate_gzip_compress_bound",
    "libdeflate_free_compressor",
    "libdeflate_alloc_decompressor", "libdeflate_alloc_decompressor_ex",
    "libdeflate_deflate_decompress", "libdeflate_deflate_decompress_ex",
    "libdeflate_zlib_decompress", "libdeflate_zlib_decompress_ex",
    "libdeflate_gzip_decompress", "libdeflate_gzip_decompress_ex",
    "libdeflate_free_decompressor",
    "libdeflate_adler32", "libdeflate_crc32",
    "libdeflate_set_memory_allocator",
]
BATCH_SYMBOLS = [
    "libdeflate_amd_device_ready", "libdeflate_amd_last_error",
    "libdeflate_amd_reload_env",
    "libdeflate_amd_compress_batch", "libdeflate_amd_decompress_batch",
    "libdeflate_amd_compress_batch_bounded",
    "libdeflate_amd_crc32_batch", "libdeflate_amd_adler32_batch",
    "libdeflate_amd_compress_batch_host",
    "libdeflate_amd_decompress_batch_host",
    "libdeflate_amd_compact_offsets_len", "libdeflate_amd_compact_batch",
    "libdeflate_amd_gzip_decompress_members",
    "libdeflate_amd_stream_stats", "libdeflate_amd_last_fanout",
    "libdeflate_amd_selfcheck",
]

_lib = None
MISSING = []


def load():
    """dlopen the HIP library (RTLD_LOCAL) and declare the signatures."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C libdeflate_amd/csrc -j` (hipcc, gfx950). "
            "libdeflate_amd has no CPU fallback.")
    # One HIP runtime per process: torch bundles its own libamdhip64 (same
    # SONAME as /opt/rocm's).  Importing torch first makes the dynamic loader
    # bind our DT_NEEDED libamdhip64.so.7 to the copy torch already mapped;
    # the other order would map two runtimes and the second one sees no GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    P, SZ = c_void_p, c_size_t
    psz = POINTER(c_size_t)

    def sig(name, restype, *argtypes):
        try:
            fn = getattr(lib, name)
        except AttributeError:
            MISSING.append(name)    # tests/test_abi.py asserts this stays empty
            return
        fn.restype = restype
        fn.argtypes = list(argtypes)

    sig("libdeflate_alloc_compressor", P, c_int)
    sig("libdeflate_alloc_compressor_ex", P, c_int, P)
    sig("libdeflate_free_compressor", None, P)
    sig("libdeflate_alloc_decompressor", P)
    sig("libdeflate_alloc_decompressor_ex", P, P)
    sig("libdeflate_free_decompressor", None, P)
    for f in ("deflate", "zlib", "gzip"):
        sig(f"libdeflate_{f}_compress", SZ, P, P, SZ, P, SZ)
        sig(f"libdeflate_{f}_compress_bound", SZ, P, SZ)
        sig(f"libdeflate_{f}_decompress", c_int, P, P, SZ, P, SZ, psz)
        sig(f"libdeflate_{f}_decompress_ex", c_int, P, P, SZ, P, SZ, psz, psz)
    sig("libdeflate_crc32", c_uint32, c_uint32, P, SZ)
    sig("libdeflate_adler32", c_uint32, c_uint32, P, SZ)
    sig("libdeflate_set_memory_allocator", None, P, P)
    sig("libdeflate_amd_device_ready", c_int)
    sig("libdeflate_amd_last_error", c_char_p)
    sig("libdeflate_amd_reload_env", None)
    sig("libdeflate_amd_crc32_batch", c_int, SZ, P, P, P, P, P, P)
    sig("libdeflate_amd_adler32_batch", c_int, SZ, P, P, P, P, P, P)
    sig("libdeflate_amd_compress_batch", c_int, P, c_int, SZ, P, P, P, P, P,
        P, P, P)
    sig("libdeflate_amd_compress_batch_bounded", c_int, P, c_int, SZ, P, P, P, P, P,
        P, P, SZ, P)
    sig("libdeflate_amd_decompress_batch", c_int, P, c_int, SZ, P, P, P, P, P,
        P, P, P, P, P)
    sig("libdeflate_amd_compress_batch_host", c_int, P, c_int, SZ, P, P, P, P,
        P)
    sig("libdeflate_amd_decompress_batch_host", c_int, P, c_int, SZ, P, P, P,
        P, P, P, P)
    sig("libdeflate_amd_gzip_decompress_members", c_int, P, P, SZ, P, SZ, psz, psz, psz)
    sig("libdeflate_amd_stream_stats", None, POINTER(c_uint64))
    sig("libdeflate_amd_last_fanout", c_size_t)
    sig("libdeflate_amd_selfcheck", c_int, POINTER(c_uint64))
    sig("libdeflate_amd_compact_offsets_len", SZ, SZ)
    sig("libdeflate_amd_compact_batch", c_int, SZ, P, P, P, P, P, P)
    _lib = lib
    return lib


def reload_env():
    """Have the library read its LDA_* tuning switches again (it reads them
    once, at load)."""
    load().libdeflate_amd_reload_env()


def last_fanout():
    """shards (devices) the calling thread's last host-pointer batch used"""
    return int(load().libdeflate_amd_last_fanout())


def selfcheck():
    """the per-device hardware self-check again -> (status, dict of counters)"""
    out = (c_uint64 * 5)()
    rc = load().libdeflate_amd_selfcheck(out)
    keys = ("lds_lanes", "lds_out_of_order", "lds_conflicts", "loads", "stale_loads")
    return rc, dict(zip(keys, (int(x) for x in out)))


def stream_stats():
    """libdeflate_amd_stream_stats: what the last single-buffer decompress
    call of this thread did (see include/libdeflate_amd.h)."""
    out = (c_uint64 * 16)()
    load().libdeflate_amd_stream_stats(out)
    keys = ("parallel", "why_not", "filter_a", "blocks_found", "chunks_planned",
            "repairs", "chunks_decoded", "bytes", "us_in", "us_find", "us_count",
            "us_decode", "us_sum", "us_out", "windows", "host_chunks")
    return dict(zip(keys, [int(v) for v in out]))


def last_error():
    return load().libdeflate_amd_last_error().decode()


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (status {rc}): {last_error()}")
