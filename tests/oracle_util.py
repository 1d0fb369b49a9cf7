"""ctypes access to the CPU checkers (oracle/).  TEST INFRASTRUCTURE ONLY -
nothing in libdeflate_amd/ imports this module."""
import ctypes
import os
import subprocess
from ctypes import POINTER, c_char_p, c_int, c_size_t, c_uint32, c_void_p

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
SZ = c_size_t


def build_oracle():
    subprocess.run(["make", "-C", ORACLE_DIR], check=True,
                   stdout=subprocess.DEVNULL)


class Oracle:
    """liboracle.so: the from-scratch CPU restatement."""

    def __init__(self, lib):
        self.lib = lib
        for f in ("deflate", "zlib", "gzip"):
            fn = getattr(lib, f"oracle_{f}_decompress")
            fn.restype = c_int
            fn.argtypes = [c_char_p, SZ, c_void_p, SZ, POINTER(SZ), POINTER(SZ)]
            fn = getattr(lib, f"oracle_{f}_compress")
            fn.restype = SZ
            fn.argtypes = [c_int, c_char_p, SZ, c_void_p, SZ]
            fn = getattr(lib, f"oracle_{f}_compress_bound")
            fn.restype = SZ
            fn.argtypes = [SZ]
        for f in ("crc32", "adler32"):
            fn = getattr(lib, f"oracle_{f}")
            fn.restype = c_uint32
            fn.argtypes = [c_uint32, c_char_p, SZ]

    def block_map(self, data, out_avail, cap=1 << 16):
        """Blocks of a raw DEFLATE stream: [(bit, out_pos, btype, final)],
        and the decoder's result code."""
        class Blk(ctypes.Structure):
            _fields_ = [("bit", ctypes.c_uint64), ("out_pos", ctypes.c_uint64),
                        ("type", c_uint32), ("pad", c_uint32)]
        fn = self.lib.oracle_deflate_block_map
        fn.restype = SZ
        fn.argtypes = [c_char_p, SZ, c_void_p, SZ, c_void_p, SZ, POINTER(c_int)]
        out = ctypes.create_string_buffer(max(out_avail, 1))
        blocks = (Blk * cap)()
        res = c_int(0)
        n = fn(bytes(data), len(data), out, out_avail, blocks, cap, ctypes.byref(res))
        return [(b.bit, b.out_pos, b.type & 3, b.type >> 2)
                for b in blocks[:min(n, cap)]], res.value

    def decompress_ex(self, fmt, data, out_avail, want_actual_out=True):
        out = ctypes.create_string_buffer(max(out_avail, 1))
        ai, ao = SZ(0), SZ(0)
        r = getattr(self.lib, f"oracle_{fmt}_decompress")(
            bytes(data), len(data), out, out_avail, ctypes.byref(ai),
            ctypes.byref(ao) if want_actual_out else None)
        nout = ao.value if want_actual_out else out_avail
        return r, ai.value, ao.value, out.raw[:nout]

    def compress(self, fmt, level, data, out_avail=None):
        if out_avail is None:
            out_avail = self.bound(fmt, len(data))
        out = ctypes.create_string_buffer(max(out_avail, 1))
        n = getattr(self.lib, f"oracle_{fmt}_compress")(
            level, bytes(data), len(data), out, out_avail)
        return out.raw[:n] if n else None

    def bound(self, fmt, n):
        return getattr(self.lib, f"oracle_{fmt}_compress_bound")(n)

    def crc32(self, data, init=0):
        return self.lib.oracle_crc32(init, bytes(data), len(data))

    def adler32(self, data, init=1):
        return self.lib.oracle_adler32(init, bytes(data), len(data))


class Ref:
    """oracle/_ref/libdeflate_ref.so: the real reference, built from its own
    sources by oracle/Makefile."""

    def __init__(self, lib):
        self.lib = lib
        lib.libdeflate_alloc_decompressor.restype = c_void_p
        lib.libdeflate_alloc_compressor.restype = c_void_p
        lib.libdeflate_alloc_compressor.argtypes = [c_int]
        lib.libdeflate_free_compressor.argtypes = [c_void_p]
        lib.libdeflate_free_decompressor.argtypes = [c_void_p]
        for f in ("deflate", "zlib", "gzip"):
            fn = getattr(lib, f"libdeflate_{f}_decompress_ex")
            fn.restype = c_int
            fn.argtypes = [c_void_p, c_char_p, SZ, c_void_p, SZ, POINTER(SZ),
                           POINTER(SZ)]
            fn = getattr(lib, f"libdeflate_{f}_compress")
            fn.restype = SZ
            fn.argtypes = [c_void_p, c_char_p, SZ, c_void_p, SZ]
            fn = getattr(lib, f"libdeflate_{f}_compress_bound")
            fn.restype = SZ
            fn.argtypes = [c_void_p, SZ]
        for f in ("crc32", "adler32"):
            fn = getattr(lib, f"libdeflate_{f}")
            fn.restype = c_uint32
            fn.argtypes = [c_uint32, c_char_p, SZ]
        self._d = c_void_p(lib.libdeflate_alloc_decompressor())
        self._c = {}

    def _comp(self, level):
        if level not in self._c:
            self._c[level] = c_void_p(self.lib.libdeflate_alloc_compressor(level))
        return self._c[level]

    def decompress_ex(self, fmt, data, out_avail, want_actual_out=True):
        out = ctypes.create_string_buffer(max(out_avail, 1))
        ai, ao = SZ(0), SZ(0)
        r = getattr(self.lib, f"libdeflate_{fmt}_decompress_ex")(
            self._d, bytes(data), len(data), out, out_avail, ctypes.byref(ai),
            ctypes.byref(ao) if want_actual_out else None)
        nout = ao.value if want_actual_out else out_avail
        return r, ai.value, ao.value, out.raw[:nout]

    def compress(self, fmt, level, data, out_avail=None):
        if out_avail is None:
            out_avail = self.bound(fmt, len(data))
        out = ctypes.create_string_buffer(max(out_avail, 1))
        n = getattr(self.lib, f"libdeflate_{fmt}_compress")(
            self._comp(level), bytes(data), len(data), out, out_avail)
        return out.raw[:n] if n else None

    def bound(self, fmt, n):
        return getattr(self.lib, f"libdeflate_{fmt}_compress_bound")(None, n)

    def crc32(self, data, init=0):
        return self.lib.libdeflate_crc32(init, bytes(data), len(data))

    def adler32(self, data, init=1):
        return self.lib.libdeflate_adler32(init, bytes(data), len(data))


def load_oracle():
    path = os.path.join(ORACLE_DIR, "liboracle.so")
    if not os.path.exists(path):
        build_oracle()
    return Oracle(ctypes.CDLL(path))


def load_ref():
    path = os.path.join(ORACLE_DIR, "_ref", "libdeflate_ref.so")
    if not os.path.exists(path):
        if os.path.isdir("/root/reference/lib"):
            build_oracle()
        if not os.path.exists(path):
            return None
    return Ref(ctypes.CDLL(path))
