"""The N>1 path on CPU: world_size 2 and 8, gloo backend.  Checks the contiguous
partition, the verdict gather and the padded payload gather that bench.py
--gpus N runs over RCCL.  (Compute stays on the GPU; here the per-rank
"results" are synthesized so the exchange logic is what is under test.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from libdeflate_amd import shard


def test_partition_covers_everything():
    for n in (0, 1, 7, 4096, 65536, 1000003):
        for world in (1, 2, 4, 8):
            spans = [shard.partition(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_total = 1001
        lo, hi = shard.partition(n_total, world, rank)
        sizes = torch.arange(lo, hi, dtype=torch.int64) % 977 + 1
        results = torch.zeros(hi - lo, dtype=torch.int32)
        if rank == 1:
            results[3] = 1          # one failed chunk on rank 1
        tot, bad = shard.gather_verdicts(sizes, results, dist, world)
        # payload: each chunk i contributes sizes[i] bytes of value i & 255
        slot = 1024
        payload = torch.zeros((hi - lo) * slot, dtype=torch.uint8)
        offs = torch.arange(hi - lo, dtype=torch.int64) * slot
        for k in range(hi - lo):
            payload[k * slot:k * slot + int(sizes[k])] = (lo + k) & 255
        seg = shard.compact(payload, offs, sizes)
        segs = shard.gather_payload(seg, dist, world)
        if rank == 0:
            whole = torch.cat(segs)
            want = torch.cat([torch.full((int(i % 977 + 1),), i & 255,
                                         dtype=torch.uint8)
                              for i in range(n_total)])
            q.put((tot, bad, bool(torch.equal(whole, want))))
    finally:
        dist.destroy_process_group()


def test_gather_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    tot, bad, ok = q.get(timeout=10)
    assert (tot, bad, ok) == (1001, 1, True)


def test_gather_world8_gloo():
    """the same exchange at the world size of a full node (8 ranks): the
    contiguous partitions add up, eight verdict vectors and eight payload
    segments arrive at rank 0 in rank order"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    tot, bad, ok = q.get(timeout=10)
    assert (tot, bad, ok) == (1001, 1, True)
