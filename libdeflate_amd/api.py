"""Host-side mirror of the reference interface, on top of the C-ABI.

`Compressor` / `Decompressor` follow libdeflate.h's object model (alloc with a
level, call `<format>_compress` / `<format>_decompress[_ex]`, free) with the
same argument meaning and result conventions, so parity tests read like the
reference's own (programs/test_trailing_bytes.c etc.).  The `*_batch`
functions take torch CUDA tensors that already live in HBM and only pass
their device pointers through; torch is plumbing (memory, streams), never
compute.
"""
import ctypes
from ctypes import c_size_t, c_void_p

import numpy as np

from . import binding
from .binding import FORMATS, check


def _buf(b):
    """ctypes pointer + length for bytes / bytearray / numpy uint8."""
    if isinstance(b, np.ndarray):
        assert b.dtype == np.uint8 and b.flags["C_CONTIGUOUS"]
        return b.ctypes.data_as(c_void_p), b.size
    if isinstance(b, (bytes, bytearray, memoryview)):
        arr = np.frombuffer(b, dtype=np.uint8)
        return arr.ctypes.data_as(c_void_p), arr.size
    raise TypeError(type(b))


class Compressor:
    """libdeflate_alloc_compressor / _free_compressor (libdeflate.h:59-160)."""

    def __init__(self, level=6):
        self._lib = binding.load()
        self.level = level
        self._h = self._lib.libdeflate_alloc_compressor(level)
        if not self._h:
            raise RuntimeError(
                f"libdeflate_alloc_compressor({level}) returned NULL: "
                f"{binding.last_error()}")

    def close(self):
        if self._h:
            self._lib.libdeflate_free_compressor(self._h)
            self._h = None

    __del__ = close

    def bound(self, fmt, n):
        return getattr(self._lib, f"libdeflate_{fmt}_compress_bound")(self._h, n)

    def compress(self, fmt, data, out_avail=None):
        """Returns the compressed bytes, or None when the reference API would
        return 0 (does not fit in out_avail)."""
        p, n = _buf(data)
        if out_avail is None:
            out_avail = self.bound(fmt, n)
        out = np.empty(max(out_avail, 1), dtype=np.uint8)
        r = getattr(self._lib, f"libdeflate_{fmt}_compress")(
            self._h, p, n, out.ctypes.data_as(c_void_p), out_avail)
        if r == 0:
            return None
        return out[:r].tobytes()

    def compress_batch(self, fmt, data, in_offsets, in_nbytes, out,
                       out_offsets, out_avail, out_nbytes, stream=None,
                       max_chunk=None):
        """Device batch: all arguments are torch CUDA tensors (uint8 data,
        int64 descriptors).  Enqueues on `stream` (torch stream or None).
        max_chunk: an upper bound of the chunk sizes, if the caller knows one
        (libdeflate_amd_compress_batch_bounded: small chunks get their own
        kernel)."""
        if max_chunk is None:
            check(self._lib.libdeflate_amd_compress_batch(
                self._h, FORMATS[fmt], in_offsets.numel(), data.data_ptr(),
                in_offsets.data_ptr(), in_nbytes.data_ptr(), out.data_ptr(),
                out_offsets.data_ptr(), out_avail.data_ptr(),
                out_nbytes.data_ptr(), _stream_ptr(stream)), "compress_batch")
        else:
            check(self._lib.libdeflate_amd_compress_batch_bounded(
                self._h, FORMATS[fmt], in_offsets.numel(), data.data_ptr(),
                in_offsets.data_ptr(), in_nbytes.data_ptr(), out.data_ptr(),
                out_offsets.data_ptr(), out_avail.data_ptr(),
                out_nbytes.data_ptr(), int(max_chunk), _stream_ptr(stream)),
                "compress_batch_bounded")

    def compress_batch_host(self, fmt, chunks, out_avail=None):
        """List of bytes -> list of compressed bytes (None where it did not
        fit), through libdeflate_amd_compress_batch_host."""
        n = len(chunks)
        arrs = [np.frombuffer(c, dtype=np.uint8) for c in chunks]
        avail = [self.bound(fmt, a.size) if out_avail is None else out_avail[i]
                 for i, a in enumerate(arrs)]
        outs = [np.empty(max(a, 1), dtype=np.uint8) for a in avail]
        inp = (c_void_p * n)(*[a.ctypes.data for a in arrs])
        inn = (c_size_t * n)(*[a.size for a in arrs])
        outp = (c_void_p * n)(*[o.ctypes.data for o in outs])
        outa = (c_size_t * n)(*avail)
        outn = (c_size_t * n)()
        check(self._lib.libdeflate_amd_compress_batch_host(
            self._h, FORMATS[fmt], n, inp, inn, outp, outa, outn),
            "compress_batch_host")
        return [outs[i][:outn[i]].tobytes() if outn[i] else None
                for i in range(n)]


class Decompressor:
    """libdeflate_alloc_decompressor / _free (libdeflate.h:181-323)."""

    def __init__(self):
        self._lib = binding.load()
        self._h = self._lib.libdeflate_alloc_decompressor()
        if not self._h:
            raise RuntimeError("libdeflate_alloc_decompressor returned NULL: "
                               + binding.last_error())

    def close(self):
        if self._h:
            self._lib.libdeflate_free_decompressor(self._h)
            self._h = None

    __del__ = close

    def decompress_ex(self, fmt, data, out_avail, want_actual_out=True):
        """-> (result, actual_in, actual_out, out_bytes) with the semantics of
        libdeflate_<fmt>_decompress_ex; want_actual_out=False passes NULL for
        actual_out_nbytes_ret (exact-fill mode)."""
        p, n = _buf(data)
        out = np.zeros(max(out_avail, 1), dtype=np.uint8)
        ai, ao = c_size_t(0), c_size_t(0)
        r = getattr(self._lib, f"libdeflate_{fmt}_decompress_ex")(
            self._h, p, n, out.ctypes.data_as(c_void_p), out_avail,
            ctypes.byref(ai), ctypes.byref(ao) if want_actual_out else None)
        nout = ao.value if want_actual_out else out_avail
        return r, ai.value, ao.value, out[:nout].tobytes()

    def decompress(self, fmt, data, out_avail, want_actual_out=True):
        p, n = _buf(data)
        out = np.zeros(max(out_avail, 1), dtype=np.uint8)
        ao = c_size_t(0)
        r = getattr(self._lib, f"libdeflate_{fmt}_decompress")(
            self._h, p, n, out.ctypes.data_as(c_void_p), out_avail,
            ctypes.byref(ao) if want_actual_out else None)
        nout = ao.value if want_actual_out else out_avail
        return r, ao.value, out[:nout].tobytes()

    def gzip_decompress_members(self, data, out_avail):
        """All members of a multi-member gzip buffer (the loop of
        programs/gzip.c:236-299 in one call) -> (result, actual_in,
        actual_out, members, bytes)."""
        p, n = _buf(data)
        out = np.zeros(max(out_avail, 1), dtype=np.uint8)
        ai, ao, nm = c_size_t(0), c_size_t(0), c_size_t(0)
        r = self._lib.libdeflate_amd_gzip_decompress_members(
            self._h, p, n, out.ctypes.data_as(c_void_p), out_avail,
            ctypes.byref(ai), ctypes.byref(ao), ctypes.byref(nm))
        return r, ai.value, ao.value, nm.value, out[:ao.value].tobytes()

    def decompress_batch(self, fmt, data, in_offsets, in_nbytes, out,
                         out_offsets, out_avail, results, actual_in=None,
                         actual_out=None, stream=None):
        check(self._lib.libdeflate_amd_decompress_batch(
            self._h, FORMATS[fmt], in_offsets.numel(), data.data_ptr(),
            in_offsets.data_ptr(), in_nbytes.data_ptr(), out.data_ptr(),
            out_offsets.data_ptr(), out_avail.data_ptr(), results.data_ptr(),
            actual_in.data_ptr() if actual_in is not None else None,
            actual_out.data_ptr() if actual_out is not None else None,
            _stream_ptr(stream)), "decompress_batch")

    def decompress_batch_host(self, fmt, chunks, out_avail,
                              want_actual_out=True):
        """-> list of (result, actual_in, actual_out, bytes)."""
        n = len(chunks)
        arrs = [np.frombuffer(c, dtype=np.uint8) for c in chunks]
        outs = [np.zeros(max(a, 1), dtype=np.uint8) for a in out_avail]
        inp = (c_void_p * n)(*[a.ctypes.data for a in arrs])
        inn = (c_size_t * n)(*[a.size for a in arrs])
        outp = (c_void_p * n)(*[o.ctypes.data for o in outs])
        outa = (c_size_t * n)(*out_avail)
        res = (ctypes.c_int32 * n)()
        ain = (c_size_t * n)()
        aout = (c_size_t * n)()
        check(self._lib.libdeflate_amd_decompress_batch_host(
            self._h, FORMATS[fmt], n, inp, inn, outp, outa, res, ain,
            aout if want_actual_out else None), "decompress_batch_host")
        r = []
        for i in range(n):
            nout = aout[i] if want_actual_out else out_avail[i]
            r.append((res[i], ain[i], aout[i],
                      outs[i][:nout].tobytes() if res[i] == 0 else b""))
        return r


def _stream_ptr(stream):
    if stream is None:
        return None
    return c_void_p(getattr(stream, "cuda_stream", stream))


def crc32(data, init=0):
    """libdeflate_crc32 (libdeflate.h:345-346) on a host buffer."""
    p, n = _buf(data)
    return binding.load().libdeflate_crc32(init, p, n)


def adler32(data, init=1):
    """libdeflate_adler32 (libdeflate.h:335-336) on a host buffer."""
    p, n = _buf(data)
    return binding.load().libdeflate_adler32(init, p, n)


def checksum_batch(kind, data, offsets, nbytes, out, init=None, stream=None):
    """Device batch CRC-32 / Adler-32; torch CUDA tensors (uint8 data, int64
    offsets/nbytes, int32 out/init holding the u32 bit patterns)."""
    lib = binding.load()
    fn = (lib.libdeflate_amd_crc32_batch if kind == "crc32"
          else lib.libdeflate_amd_adler32_batch)
    check(fn(offsets.numel(), data.data_ptr(), offsets.data_ptr(),
             nbytes.data_ptr(), init.data_ptr() if init is not None else None,
             out.data_ptr(), _stream_ptr(stream)), kind + "_batch")


def compact_batch(data, offsets, nbytes, stream=None):
    """Device-side compaction (libdeflate_amd_compact_batch): the used part of
    every slot back to back.  torch CUDA tensors in; returns (packed uint8
    tensor sized for the worst case, int64 offsets[n + 1] - exclusive prefix
    sums of nbytes, offsets[n] = total).  Only enqueues; slice `packed` with
    int(offsets[n]) after synchronising."""
    import torch
    lib = binding.load()
    n = offsets.numel()
    cap = int(lib.libdeflate_amd_compact_offsets_len(n))
    out_off = torch.zeros(cap, dtype=torch.int64, device=data.device)
    packed = torch.empty(data.numel(), dtype=torch.uint8, device=data.device)
    check(lib.libdeflate_amd_compact_batch(
        n, data.data_ptr(), offsets.data_ptr(), nbytes.data_ptr(),
        packed.data_ptr(), out_off.data_ptr(), _stream_ptr(stream)),
        "compact_batch")
    return packed, out_off[:n + 1]
