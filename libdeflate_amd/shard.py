"""Sharding a batch of independent chunks over the GPUs of one node.

Chunks are self-contained streams (fresh match-finder state and fresh Huffman
codes per call: lib/deflate_compress.c:2616,2630; decoder state on the stack:
lib/decompress_template.h:50-70), so the data path needs no collective: rank g
of G owns the contiguous index range [g*N/G, (g+1)*N/G) (contiguous keeps the
gather a concatenation and preserves input order).  The only exchange is the
final gather to the root, over RCCL (torch.distributed backend "nccl") on
GPUs, gloo in the CPU tests:

  gather_verdicts   fixed-size per-chunk metadata (compressed size, status)
  gather_payload    the variable-length compressed bytes: every rank sends its
                    compacted segment to the root, point to point (grouped
                    send / recv: xGMI is point-to-point, and only the root
                    needs the bytes)

Under gloo the collectives run on host copies of the tensors, so that the
same control flow can be exercised by several ranks sharing one GPU.

A caller without a process group passes dist=None and gets its own data back.
A process group of ONE rank still goes through every collective (all_reduce,
gather; no peer, so no send / recv): `bench.py --gpus 1 --force-dist` and
tests/test_rccl_gpu.py run the RCCL code path that way on a single-GPU box.
"""
import torch


def _coll(t, dist):
    """The tensor a collective works on: `t` itself under RCCL, a host copy
    under gloo."""
    if t.is_cuda and dist.get_backend() == "gloo":
        return t.cpu()
    return t


def partition(n_chunks, world, rank):
    """Contiguous shard [lo, hi) of rank `rank`."""
    lo = rank * n_chunks // world
    hi = (rank + 1) * n_chunks // world
    return lo, hi


def gather_verdicts(sizes, results, dist, world):
    """Gather (size, status) of every chunk to rank 0.  Returns
    (total_chunks, n_failed) on rank 0 and (local count, local failed)
    elsewhere.  sizes: int64[n]; results: int32[n]."""
    packed = torch.stack([sizes, results.to(torch.int64)], dim=1).contiguous()
    if dist is None:
        return packed.shape[0], int((packed[:, 1] != 0).sum())
    packed = _coll(packed, dist)
    rank = dist.get_rank()
    # shards may differ by one chunk: pad to the largest, first row = count
    cnt = torch.tensor([packed.shape[0]], dtype=torch.int64, device=packed.device)
    mx = cnt.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    padded = torch.zeros((int(mx.item()) + 1, 2), dtype=torch.int64,
                         device=packed.device)
    padded[0, 0] = packed.shape[0]
    padded[1:1 + packed.shape[0]] = packed
    bufs = [torch.empty_like(padded) for _ in range(world)] if rank == 0 else None
    dist.gather(padded, bufs, dst=0)
    if rank == 0:
        allv = torch.cat([b[1:1 + int(b[0, 0].item())] for b in bufs], dim=0)
        return allv.shape[0], int((allv[:, 1] != 0).sum())
    return packed.shape[0], int((packed[:, 1] != 0).sum())


def compact(payload, offsets, sizes):
    """Concatenate the used part of each output slot into one uint8 tensor on
    payload's device.  On a GPU this is the device-side compaction of the
    C-ABI (prefix sum of the sizes + one gather-copy kernel, no per-chunk host
    work); CPU tensors (the gloo tests) are concatenated with torch."""
    if payload.is_cuda:
        from . import api
        packed, off = api.compact_batch(payload, offsets, sizes)
        return packed[:int(off[-1].item())]
    parts = [payload[int(o):int(o) + int(s)] for o, s in
             zip(offsets.tolist(), sizes.tolist())]
    return torch.cat(parts) if parts else payload[:0]


def gather_payload(segment, dist, world):
    """All ranks contribute one compacted uint8 segment; rank 0 gets the list
    of segments in rank order (others get None).  The lengths are gathered
    first; then every other rank sends its bytes to the root and the root
    posts the matching receives (one transfer per peer, nothing travels to a
    rank that does not need it - SURVEY.md 8(e): grouped ncclSend/ncclRecv)."""
    if dist is None:
        return [segment]
    segment = _coll(segment.contiguous(), dist)
    rank = dist.get_rank()
    n = torch.tensor([segment.numel()], dtype=torch.int64, device=segment.device)
    lens = [torch.zeros_like(n) for _ in range(world)] if rank == 0 else None
    dist.gather(n, lens, dst=0)
    if rank != 0:
        if segment.numel():
            dist.send(segment, dst=0)
        return None
    out, reqs = [segment], []
    for r in range(1, world):
        buf = torch.empty(int(lens[r].item()), dtype=torch.uint8, device=segment.device)
        out.append(buf)
        if buf.numel():
            reqs.append(dist.irecv(buf, src=r))
    for q in reqs:
        q.wait()
    return out
