"""RCCL executed: the N > 1 path's process group and every
collective it uses, run at world size 1 on the single GPU of the test box with
the "nccl" backend (= RCCL on ROCm) and DEVICE tensors.  No peer exists, so
nothing crosses xGMI; what this proves is that librccl loads, that a
communicator comes up on the device the library uses, and that the gather
code of libdeflate_amd/shard.py runs through RCCL on device tensors (under
gloo it ran on host copies only).  Each case is a subprocess under a timeout:
a collective that hangs must fail the test, not the suite."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import os, sys, json
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
from libdeflate_amd import shard
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", world_size=1, rank=0, device_id=dev)
out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
dist.barrier()
# gather_verdicts: all_reduce(MAX) + gather of int64 device tensors
sizes = torch.arange(1000, 1100, dtype=torch.int64, device=dev)
res = torch.zeros(100, dtype=torch.int32, device=dev)
res[7] = 1
res[41] = 3
out["verdicts"] = list(shard.gather_verdicts(sizes, res, dist, 1))
# gather_payload: gather of the lengths (device), the root's own segment
seg = torch.arange(5000, dtype=torch.int64, device=dev).to(torch.uint8)
segs = shard.gather_payload(seg, dist, 1)
out["payload_ok"] = len(segs) == 1 and segs[0].is_cuda and bool(torch.equal(segs[0], seg))
# the point-to-point half (ncclSend / ncclRecv) has no peer at world 1: a grouped
# send + receive to the own rank pushes device bytes through it all the same
rx = torch.zeros_like(seg)
try:
    ops = [dist.P2POp(dist.isend, seg, 0), dist.P2POp(dist.irecv, rx, 0)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    torch.cuda.synchronize()
    out["self_p2p"] = bool(torch.equal(rx, seg))
except Exception as e:      # not every RCCL build allows a send to self
    out["self_p2p"] = "unsupported: " + type(e).__name__
t = torch.tensor([3.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
out["all_reduce"] = float(t.item())
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
"""


def _env():
    env = dict(os.environ)
    env["NCCL_DEBUG"] = "VERSION"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    return env


def test_shard_collectives_run_through_rccl_on_device_tensors():
    r = subprocess.run([sys.executable, "-c", _SCRIPT % {"root": ROOT}], env=_env(),
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    log = r.stdout + r.stderr
    assert r.returncode == 0, log[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    print(out)
    print([l for l in log.splitlines() if "version" in l.lower()][:3])
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["verdicts"] == [100, 2]
    assert out["payload_ok"] is True
    assert out["all_reduce"] == 3.5
    assert out["self_p2p"] is True or str(out["self_p2p"]).startswith("unsupported")
    # NCCL_DEBUG=VERSION: the library that ran says which one it is
    assert any(("RCCL" in l or "NCCL" in l) and "version" in l.lower()
               for l in log.splitlines()), log[-2000:]
    # ... and it is mapped from torch's ROCm build, not a CUDA NCCL
    maps = subprocess.run([sys.executable, "-c",
                           "import torch, torch.distributed as d, os; "
                           "print([l.split()[-1] for l in open('/proc/self/maps') "
                           "if 'rccl' in l.lower() or 'nccl' in l.lower()][:1])"],
                          capture_output=True, text=True, timeout=300)
    print("mapped:", maps.stdout.strip())


def test_bench_runs_its_n_gpu_path_at_world_1_over_rccl():
    """bench.py --force-dist: init_process_group("nccl"), the barriers, the
    max-over-ranks all_reduce of the timing and the verdict gather INSIDE the
    timed step, all on the single leased GPU."""
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--backend", "nccl",
                        "--force-dist", "--steps", "3", "--warmup", "1", "--no-cpu",
                        "--configs", "headline", "--chunks", "512"],
                       env=_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    log = r.stdout + r.stderr
    assert r.returncode == 0, log[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print({k: line[k] for k in ("value", "n_gpus", "collectives", "verdicts")})
    assert line["collectives"] == {"backend": "nccl", "world": 1, "forced_at_world_1": True}
    assert line["n_gpus"] == 1 and line["verdicts"] == {"chunks": 512, "failed": 0}
    assert line["verified"].startswith("round trip byte-exact")
