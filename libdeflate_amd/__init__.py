"""libdeflate_amd - MI355X-native whole-buffer DEFLATE engine.

The product is libdeflate_amd.so (HIP kernels + the libdeflate.h-compatible
C-ABI, see include/libdeflate_amd.h).  This package is the thin host layer:
`binding` declares the C signatures for ctypes, `api` mirrors the reference's
compressor / decompressor objects and adds torch-tensor batch helpers, and
`shard` partitions a batch over the GPUs of one node.
"""
from . import binding  # noqa: F401

__all__ = ["binding"]
