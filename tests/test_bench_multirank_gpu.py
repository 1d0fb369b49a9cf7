"""bench.py's N > 1 control flow on one GPU: `python bench.py --gpus 2`
re-launches itself under torch.distributed.run exactly as the driver's
multi-GPU tier does; with `--backend gloo --one-device` both ranks share
cuda:0 and the collectives (barrier, max-over-ranks timing, verdict gather,
all_sum) run on host tensors.  What is checked is what a SCALE run depends on:
the re-launch, rank-0 JSON assembly, the strong-scaling partitions summing to
the totals, and every `verified` string.  (RCCL itself is not exercised here:
that needs one GPU per rank.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _run(args):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args,
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]      # ONE line, from rank 0
    return json.loads(lines[0])


def test_bench_two_ranks_one_device():
    streams, blocks, chunks = 4096, 65536, 512
    line = _run(["--gpus", "2", "--backend", "gloo", "--one-device", "--configs", "all",
                 "--streams", str(streams), "--blocks", str(blocks),
                 "--chunks", str(chunks), "--steps", "2", "--warmup", "1", "--no-cpu"])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    assert line["scaling"] == "weak" and line["unit"] == "MB/s"
    # weak scaling: every rank has its own `chunks`; verdicts of both at rank 0
    assert line["verdicts"] == {"chunks": 2 * chunks, "failed": 0}
    assert line["value"] > 0 and line["roofline"]["frac"] > 0
    cfg = line["configs"]
    c1 = cfg["configs[1]"]
    assert c1["scaling"] == "weak" and c1["n_gpus"] == 2
    assert c1["chunks_total"] == 2 * chunks
    assert c1["verdicts"] == {"chunks": 2 * chunks, "failed": 0}
    assert "byte-exact" in c1["verified"]
    c3 = cfg["configs[3]"]
    assert c3["scaling"] == "strong" and c3["n_gpus"] == 2
    assert c3["streams_total"] == streams          # the partitions sum to the total
    assert c3["verdicts"] == {"chunks": streams, "failed": 0}
    assert c3["verified"].startswith(f"all {streams * 65536} output bytes equal")
    c4 = cfg["configs[4]"]
    assert c4["scaling"] == "strong" and c4["n_gpus"] == 2
    assert c4["chunks_total"] == blocks
    assert c4["verdicts"] == {"chunks": blocks, "failed": 0}
    assert "byte-exact" in c4["verified"]
    assert "end_to_end" in line                     # rank 0 only


def test_bench_eight_ranks_one_device():
    """the world size of a full node: eight ranks share cuda:0 (small batches,
    so that it stays under a minute); n_gpus 8, eight partitions that add up,
    eight verdict vectors at rank 0"""
    streams, blocks, chunks = 1024, 16384, 64
    line = _run(["--gpus", "8", "--backend", "gloo", "--one-device", "--configs", "all",
                 "--streams", str(streams), "--blocks", str(blocks),
                 "--chunks", str(chunks), "--steps", "2", "--warmup", "1", "--no-cpu"])
    assert line["n_gpus"] == 8 and line["scaling"] == "weak"
    assert line["verdicts"] == {"chunks": 8 * chunks, "failed": 0}
    cfg = line["configs"]
    assert cfg["configs[1]"]["chunks_total"] == 8 * chunks
    assert cfg["configs[1]"]["verdicts"] == {"chunks": 8 * chunks, "failed": 0}
    c3 = cfg["configs[3]"]
    assert c3["n_gpus"] == 8 and c3["streams_total"] == streams
    assert c3["verdicts"] == {"chunks": streams, "failed": 0}
    assert c3["verified"].startswith(f"all {streams * 65536} output bytes equal")
    c4 = cfg["configs[4]"]
    assert c4["n_gpus"] == 8 and c4["chunks_total"] == blocks
    assert c4["verdicts"] == {"chunks": blocks, "failed": 0}


def test_bench_single_rank_line_shape():
    line = _run(["--configs", "headline", "--chunks", "256", "--steps", "2",
                 "--warmup", "1", "--no-cpu"])
    assert line["n_gpus"] == 1 and line["verdicts"]["failed"] == 0
    for key in ("metric", "value", "unit", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in line
    assert line["config"]["workload"].startswith("configs[2]")
