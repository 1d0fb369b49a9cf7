"""Devices: an object runs on the device it was allocated on whatever the
calling thread has current, and one host-pointer batch can be spread over the
GPUs of a node from ONE object (LDA_DEVICES, csrc/host_fanout.hip) - contiguous
shards, an object and a host thread per shard, results in host order, no
collective.  On a one-GPU box the plan is one shard: the
fan-out machinery itself (plan, per-shard objects, threads, results in place)
is exercised with LDA_FANOUT_OVERSUB, which lets several shards share the GPU."""
import zlib

import pytest

from libdeflate_amd import binding
from tests import datagen

pytestmark = pytest.mark.gpu


def _chunks(n, size, seed):
    return [datagen.chunk(i, size, seed) for i in range(n)]


def test_one_visible_device_is_one_shard(monkeypatch):
    import torch
    from libdeflate_amd import api
    monkeypatch.setenv("LDA_DEVICES", "all")
    binding.reload_env()
    # a shard is at least 1 MiB (fanout_plan): 2 MiB per visible device, so that
    # an 8-GPU node gets its 8 shards (64 chunks gave at most 4)
    ndev = torch.cuda.device_count()
    chunks = _chunks(32 * max(2, ndev), 65536, 0xD0)
    mib = sum(map(len, chunks)) >> 20
    c, d = api.Compressor(6), api.Decompressor()
    comp = c.compress_batch_host("gzip", chunks)
    assert binding.last_fanout() == min(ndev, 16, mib)
    got = d.decompress_batch_host("gzip", comp, [len(x) for x in chunks])
    assert binding.last_fanout() == min(ndev, 16, mib)
    assert all(g[0] == 0 and g[3] == x for g, x in zip(got, chunks))
    c.close(); d.close()


@pytest.mark.parametrize("shards", [2, 4, 8])
def test_fanout_shards_share_the_device(monkeypatch, oracle, shards):
    """the multi-device path on whatever devices there are: `shards` objects
    and threads, the streams byte for byte those of the one-shard call"""
    from libdeflate_amd import api
    chunks = _chunks(160, 65536, 0xD1) + [b"", b"x", datagen.text_chunk(300000, 5)]
    c, d = api.Compressor(6), api.Decompressor()
    monkeypatch.delenv("LDA_DEVICES", raising=False)
    binding.reload_env()
    want = c.compress_batch_host("zlib", chunks)
    assert binding.last_fanout() == 1
    monkeypatch.setenv("LDA_DEVICES", str(shards))
    monkeypatch.setenv("LDA_FANOUT_OVERSUB", "1")
    binding.reload_env()
    comp = c.compress_batch_host("zlib", chunks)
    assert binding.last_fanout() == shards
    assert comp == want                      # same kernels, same bytes, host order
    for x, z in zip(chunks, comp):
        assert zlib.decompress(z) == x
    # damaged and short-output streams keep their own result codes in place
    bad = list(comp)
    bad[5] = bad[5][:len(bad[5]) // 2]
    bad[70] = bad[70][:40] + bytes([bad[70][40] ^ 0x10]) + bad[70][41:]
    avail = [len(x) for x in chunks]
    avail[33] -= 1
    got = d.decompress_batch_host("zlib", bad, avail)
    assert binding.last_fanout() == shards
    for i, (g, x) in enumerate(zip(got, chunks)):
        exp = oracle.decompress_ex("zlib", bad[i], avail[i])
        assert g[0] == exp[0], (i, g[:3], exp[:3])
        if exp[0] == 0:
            assert g[1] == exp[1] and g[3] == x
    # a batch too small to share stays on the object's own device
    few = c.compress_batch_host("zlib", chunks[:3])
    assert binding.last_fanout() == 1 and few == want[:3]
    c.close(); d.close()


def test_object_runs_on_its_own_device(monkeypatch):
    """an object allocated on device A, called with device B current: the
    call runs on A and device B is current again afterwards"""
    import torch
    from libdeflate_amd import api
    if torch.cuda.device_count() < 2:
        # one GPU: the guard's fast path (the device is already current);
        # still make sure a call leaves the current device alone
        c = api.Compressor(1)
        before = torch.cuda.current_device()
        z = c.compress("gzip", b"device affinity " * 1000)
        assert zlib.decompress(z, 31) == b"device affinity " * 1000
        assert torch.cuda.current_device() == before
        c.close()
        pytest.skip("one visible GPU: the cross-device case needs two")
    data = datagen.text_chunk(200000, 9)
    with torch.cuda.device(1):
        c, d = api.Compressor(6), api.Decompressor()
    with torch.cuda.device(0):
        z = c.compress("gzip", data)
        assert torch.cuda.current_device() == 0
        r = d.decompress_ex("gzip", z, len(data))
        assert torch.cuda.current_device() == 0
    assert r == (0, len(z), len(data), data)
    c.close(); d.close()
