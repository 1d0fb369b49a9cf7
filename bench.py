#!/usr/bin/env python3
"""bench.py - the headline metric of BASELINE.json on MI355X.

    metric   compress+decompress MB/s, 64 KiB chunks, level 6
    workload BASELINE.json configs[2]: a batch of 4096 x 64 KiB independent
             gzip buffers, level 6 (hash-chain lazy parse) + CRC-32, per GPU.

One "step" = one pass of the hot path over one batch that is already resident
in HBM: compress the whole batch (gzip, level 6, CRC-32 included), then
decompress the whole batch (CRC-32 + ISIZE verified).  `value` is the
round-trip throughput: uncompressed batch bytes / step time, MB = 1e6 bytes
(programs/test_util.c:197-200), aggregated over all ranks; the separate
compress and decompress rates are reported alongside.

Multi-GPU: `python bench.py --gpus N` launches N ranks itself (it re-executes
under torch.distributed.run when WORLD_SIZE is not set); under an external
launcher it reads RANK/LOCAL_RANK/WORLD_SIZE.  Chunks are independent, so
every rank owns its own shard (no data-path collective); the only exchange is
the final gather of the per-chunk (size, status) verdicts to rank 0 over
RCCL, inside the timed region.

The JSON line also carries
  roofline      for the dominant kernel (the LZ77+Huffman compress kernel):
                algorithmic bytes (U read + C written) per launch / its
                average duration, HIP events on the launch stream, vs 8 TB/s.
  cpu_baseline  the real reference (oracle/_ref, built from /root/reference's
                sources) timed by a C pthread harness (oracle/cpu_bench.c) on
                this box's host cores, T = 1 and T = all, on a bounded sample
                of the same batch.
  configs       the other BASELINE.json configurations, each with its own
                ms / MB/s / roofline fraction (see extra_*()): configs[1]
                (level 1 raw), configs[3] (decompress-only, 65 536 streams
                compressed by the reference, contiguous shard per rank,
                every byte compared), configs[4] (1 Mi x 4 KiB zlib level 9).
  end_to_end    the headline batch starting and ending in pinned host memory
                (H2D + kernels + D2H); never `value`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHUNK = 65536
CHUNKS_PER_GPU = 4096
LEVEL = 6
FMT = "gzip"
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def relaunch(a):
    """`python bench.py --gpus N` without a launcher: start N ranks on this
    node (one per GPU) the way the driver does for N > 1."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def pmc_kernel(what, kernel):
    """Per-launch PMC counters of `kernel` from the newest committed profile of
    workload `what` (profiles/*pmc_<what>.json, written by tools/prof_pmc.sh:
    separate counter passes over that workload) -> (dict, relative path) or
    ({}, None).  NOT measured in this run; every use says so in the line."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*pmc_{what}.json")))
    if not files:
        return {}, None
    return (json.load(open(files[-1])).get(kernel, {}), os.path.relpath(files[-1], ROOT))


def kernel_src_digest():
    """sha256 prefix over the kernel sources and their build flags: which build
    a committed PMC profile belongs to (tools/prof_pmc.sh records it in the
    profile's `_workload`).  A profile of another build gives no issue_frac."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "libdeflate_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) +
                    [os.path.join(d, "Makefile")]):
        h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return h.hexdigest()[:16]


def profile_is_of_this_build(path):
    """(True / False / None = the profile does not say) - ADVICE r5: instruction
    counts of another build must not be combined with this run's launch time."""
    try:
        w = json.load(open(path)).get("_workload", {})
    except (OSError, ValueError):
        return None
    sha = w.get("kernel_src_sha16")
    return None if sha is None else sha == kernel_src_digest()


def issue_roof(what, kernel, t_launch, launches=1):
    """The roof the compress / decompress kernels actually run against: VALU
    issue.  A wave64 VALU instruction occupies its SIMD for 4 cycles, an
    MI355X has 1024 SIMDs: issue_frac = VALU wave-instructions per launch x 4 /
    (1024 x 2.4 GHz x launch time), instructions from the committed PMC pass
    (static), time from this run.  None when no profile of the workload is
    committed."""
    k, src = pmc_kernel(what, kernel)
    v = k.get("SQ_INSTS_VALU_per_launch")
    if not v or not t_launch:
        return None
    same = profile_is_of_this_build(os.path.join(ROOT, src))
    out = {"kernel": kernel,
           "valu_wave_insts_per_launch": int(v),
           # only a profile of THIS build may be combined with this run's time
           "issue_frac": round(v * launches * 4 / (1024 * 2.4e9 * t_launch), 3) if same else None,
           "profile_of_this_build": same,
           "source": f"{src} (static: PMC passes of tools/prof_pmc.sh, not this run) "
                     "/ this run's launch time; nominal 2.4 GHz"}
    if "FETCH_SIZE_per_launch" in k and "WRITE_SIZE_per_launch" in k:
        # FETCH_SIZE doubled per MI355X_MICROARCH.md, WRITE_SIZE as is; KiB
        out["traffic"] = int(launches * (2 * k["FETCH_SIZE_per_launch"] +
                                         k["WRITE_SIZE_per_launch"]) * 1024)
    if k.get("SQ_WAVE_CYCLES_per_launch") and k.get("SQ_WAIT_ANY_per_launch"):
        out["wave_time_waiting"] = round(k["SQ_WAIT_ANY_per_launch"] /
                                         k["SQ_WAVE_CYCLES_per_launch"], 3)
    return out


def pmc_static():
    """PMC figures of the compress kernel from a committed profile
    (tools/prof_pmc.sh bench -> profiles/*pmc_bench*.json: counter passes over
    this very batch): NOT measured in this run, and labelled as such in the
    line.  FETCH_SIZE is doubled per
    MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read),
    WRITE_SIZE taken as is; both in KiB."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_bench*.json"))) or \
        sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_deflate*.json")))
    if not files:
        return None, None, None
    src = os.path.relpath(files[-1], ROOT)
    k = json.load(open(files[-1])).get("lda_deflate_batch_kernel", {})
    traffic = None
    if "FETCH_SIZE_per_launch" in k and "WRITE_SIZE_per_launch" in k:
        traffic = int(2 * k["FETCH_SIZE_per_launch"] * 1024 +
                      k["WRITE_SIZE_per_launch"] * 1024)
    valu = k.get("SQ_INSTS_VALU_per_launch")
    if not profile_is_of_this_build(files[-1]):
        valu = None  # another build's instruction count says nothing about this run
    return traffic, valu, src


def inflate_static(t_dec):
    """The same for the decompress kernel (profiles/*pmc_inflate*.json)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_bench*.json"))) or \
        sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_inflate*.json")))
    if not files:
        return {}
    k = json.load(open(files[-1])).get("lda_inflate_wave_kernel", {})
    out = {"traffic_source": os.path.relpath(files[-1], ROOT) +
           " (static: PMC passes of tools/prof_pmc.sh, not this run)"}
    if "FETCH_SIZE_per_launch" in k and "WRITE_SIZE_per_launch" in k:
        out["traffic"] = int(2 * k["FETCH_SIZE_per_launch"] * 1024 +
                             k["WRITE_SIZE_per_launch"] * 1024)
    if k.get("SQ_INSTS_VALU_per_launch") and profile_is_of_this_build(files[-1]):
        v = k["SQ_INSTS_VALU_per_launch"]
        out["issue"] = {"valu_wave_insts_per_launch": int(v),
                        "simd_issue_frac": round(v * 4 / (1024 * 2.4e9 * t_dec), 3),
                        "note": "of the kernel's duration; while its waves are resident "
                                "(SQ_WAVE_CYCLES) the VALU is busy nearly all the time"}
    return out


def usable_cores():
    """Cores this process may actually run on: the affinity mask, cut by a
    cgroup CPU quota if there is one (a container often shows the machine's
    CPU count while being allowed a fraction of it)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    cores = min(cores, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    cores = min(cores, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return cores


def cpu_info():
    model, cores = "unknown", usable_cores()
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model, cores


def _gen_range(args):
    lo, hi, size, seed, mix4k = args
    from tests import datagen
    mix = datagen.MIX4K if mix4k else datagen.MIX64K
    return b"".join(datagen.chunk(i, size, seed, mix) for i in range(lo, hi))


def gen_chunks(count, size, seed, mix4k=False):
    """`count` DISTINCT chunks (chunk i: kind i mod 8, seed + i; SURVEY.md
    8(d)), generated by the host's cores side by side and kept in /tmp for
    the next run."""
    import concurrent.futures as cf
    path = f"/tmp/lda_bench_{seed:x}_{count}_{size}_{int(mix4k)}.bin"
    blob = None
    if os.path.exists(path) and os.path.getsize(path) == count * size:
        blob = open(path, "rb").read()
    else:
        nw = max(1, min(usable_cores(), 32))
        per = max(1, (count + 4 * nw - 1) // (4 * nw))
        jobs = [(lo, min(lo + per, count), size, seed, mix4k) for lo in range(0, count, per)]
        try:
            with cf.ProcessPoolExecutor(nw) as ex:
                blob = b"".join(ex.map(_gen_range, jobs))
        except (OSError, RuntimeError):
            blob = b"".join(_gen_range(j) for j in jobs)
        try:
            with open(path + ".tmp", "wb") as f:
                f.write(blob)
            os.replace(path + ".tmp", path)
        except OSError:
            pass
    return [blob[i * size:(i + 1) * size] for i in range(count)]


def ref_verify(torch, ref, fmt, chunks, comp, c_off, c_n, first, distinct):
    """Every DISTINCT compressed stream of a batch decoded by the real
    reference (oracle/_ref), outside the timed region: chunk i of the batch is
    chunks[(first + i) % len(chunks)], so the first `distinct` streams cover
    them all.  Returns how many were checked."""
    import ctypes
    import numpy as np
    k = min(distinct, c_n.numel())
    sizes = c_n[:k].cpu().numpy()
    offs = c_off[:k].cpu().numpy()
    hi = int(offs[k - 1] + sizes[k - 1])
    host = comp[:hi].cpu().numpy()
    size = len(chunks[0])
    out = ctypes.create_string_buffer(size)
    ai, ao = ctypes.c_size_t(0), ctypes.c_size_t(0)
    fn = getattr(ref.lib, f"libdeflate_{fmt}_decompress_ex")
    for i in range(k):
        z = host[int(offs[i]):int(offs[i]) + int(sizes[i])]
        r = fn(ref._d, z.ctypes.data_as(ctypes.c_char_p), int(sizes[i]), out, size,
               ctypes.byref(ai), ctypes.byref(ao))
        assert r == 0 and ai.value == int(sizes[i]) and ao.value == size, \
            ("the reference rejects stream", i, r)
        assert out.raw == chunks[(first + i) % len(chunks)], ("reference decodes other bytes", i)
    return k


def cpu_baseline(chunks, fmt, level, mode="rt", only_t1=False):
    """Reference libdeflate (oracle/_ref) on this box's host cores through
    oracle/cpu_bench.c (pthreads, one compressor + decompressor per thread,
    clock_gettime, best of 3 after a warm-up): T = all cores, >= 0.5 s of
    work per thread and pass, and T = 1.  mode: rt (compress + decompress,
    `value` = round trip), c, d (decompress only: `value` = decompress)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "cpu_bench")
    if not os.path.exists(exe):
        return None
    model, cores = cpu_info()
    size = len(chunks[0])
    chunks = chunks[:max(1, min(len(chunks), (32 << 20) // size))]   # the sample's distinct chunks
    out = {}
    with tempfile.NamedTemporaryFile(dir="/tmp", suffix=".bin") as f:
        # >= 0.5 s of work per thread and pass (the reference does ~125 MB/s
        # per core of an EPYC 9575F at level 6, round trip ~110 MB/s)
        per_thread = max(8, int(72e6 // size)) if level >= 5 else max(16, int(160e6 // size))
        if size > (1 << 20):
            per_thread = max(2, int(72e6 // size))
        count_all = per_thread * cores      # chunk i = file chunk i mod len(chunks)
        for c in chunks:
            f.write(c)
        f.flush()
        runs = [("all", cores, count_all), ("t1", 1, min(count_all, per_thread * 2))]
        if only_t1:
            runs = [("all", 1, per_thread)]
        for key, t, cnt in runs:
            r = subprocess.run([exe, f.name, str(size), str(cnt), fmt, str(level),
                                str(t), "3", mode], capture_output=True, text=True,
                               timeout=300)
            if r.returncode != 0:
                return {"error": (r.stderr or r.stdout)[-200:]}
            out[key] = json.loads(r.stdout)
    a, t1 = out["all"], out.get("t1", out["all"])
    key = {"rt": "MBps", "c": "compress_MBps", "d": "decompress_MBps"}[mode]
    what = {"rt": "compress+decompress round trip", "c": "compress only",
            "d": "decompress only (streams compressed by the reference itself)"}[mode]
    return {"value": a[key], "unit": "MB/s", "cores": a["threads"],
            "kind": "reference",
            "sample": f"{a['chunks']} chunks drawn from {len(chunks)} distinct {size}-byte "
                      f"chunks of the same batch ({fmt} level {level} {what}, statically "
                      f"partitioned over {a['threads']} threads, best of 3 passes "
                      f"after a warm-up, {a['wall_s']:.2f} s per pass)",
            "compress_MBps": a["compress_MBps"] if mode != "d" else None,
            "decompress_MBps": a["decompress_MBps"] if mode != "c" else None,
            "t1": {"value": t1[key],
                   "compress_MBps": t1["compress_MBps"] if mode != "d" else None,
                   "decompress_MBps": t1["decompress_MBps"] if mode != "c" else None,
                   "chunks": t1["chunks"]},
            "cpu_model": model, "host_cores": cores,
            "host_cpus_online": os.cpu_count(),
            "harness": "oracle/cpu_bench.c (pthreads, clock_gettime)"}


class Timer:
    """HIP events on the launch stream + wall clock bracketed by barriers."""

    def __init__(self, torch, dist, dev, stream, steps, nmarks):
        self.torch, self.dist, self.dev, self.stream = torch, dist, dev, stream
        self.ev = [[torch.cuda.Event(enable_timing=True) for _ in range(nmarks)]
                   for _ in range(steps)]

    def run(self, step, steps, warmup):
        torch, dist = self.torch, self.dist
        for _ in range(warmup):
            step(None)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        last = None
        for i in range(steps):
            last = step(self.ev[i])
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if dist:
            t = torch.tensor([elapsed], dtype=torch.float64,
                             device=coll_device(dist, self.dev))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, last

    def span(self, a, b):
        import numpy as np
        return float(np.mean([e[a].elapsed_time(e[b]) for e in self.ev])) / 1e3


def coll_device(dist, dev):
    """Where a collective's tensors live: the GPU under RCCL ("nccl"), the
    host under gloo (the 2-ranks-on-one-GPU test of the N > 1 control flow)."""
    return "cpu" if dist.get_backend() == "gloo" else dev


def all_sum(torch, dist, dev, *vals):
    if not dist:
        return vals
    t = torch.tensor(list(vals), dtype=torch.float64, device=coll_device(dist, dev))
    dist.all_reduce(t)
    return tuple(float(x) for x in t.tolist())


def device_batch(torch, dev, chunks, count, first=0):
    """`count` chunks on the device, chunk i = chunks[(first + i) % len]
    (independent streams either way; tiling only bounds host-side generation
    time).  Returns (data u8[count * size], offsets, nbytes)."""
    size = len(chunks[0])
    d = len(chunks)
    base = torch.frombuffer(bytearray(b"".join(chunks)), dtype=torch.uint8).to(dev)
    idx = (torch.arange(count, device=dev) + first) % d
    data = base.view(d, size)[idx].reshape(-1).contiguous()
    offs = torch.arange(count, dtype=torch.int64, device=dev) * size
    nb = torch.full((count,), size, dtype=torch.int64, device=dev)
    return data, offs, nb


def roundtrip_config(torch, dist, world, rank, dev, stream, api, shard, chunks,
                     count, first, fmt, level, steps, warmup):
    """compress then decompress `count` device-resident chunks; verifies the
    round trip byte for byte after the timed region."""
    size = len(chunks[0])
    data, in_off, in_n = device_batch(torch, dev, chunks, count, first)
    comp_c, dec = api.Compressor(level), api.Decompressor()
    bound = (comp_c.bound(fmt, size) + 15) // 16 * 16
    comp = torch.zeros(count * bound, dtype=torch.uint8, device=dev)
    c_off = torch.arange(count, dtype=torch.int64, device=dev) * bound
    c_av = torch.full((count,), bound, dtype=torch.int64, device=dev)
    c_n = torch.zeros(count, dtype=torch.int64, device=dev)
    out = torch.zeros(count * size, dtype=torch.uint8, device=dev)
    res = torch.full((count,), -1, dtype=torch.int32, device=dev)
    tm = Timer(torch, dist, dev, stream, steps, 3)

    def step(ev):
        if ev:
            ev[0].record(stream)
        comp_c.compress_batch(fmt, data, in_off, in_n, comp, c_off, c_av, c_n,
                              stream=stream, max_chunk=size)
        if ev:
            ev[1].record(stream)
        dec.decompress_batch(fmt, comp, c_off, c_n, out, in_off, in_n, res,
                             stream=stream)
        if ev:
            ev[2].record(stream)
        # final gather of the per-chunk verdicts (the only exchange step)
        return shard.gather_verdicts(c_n, res, dist, world)

    elapsed, verdict = tm.run(step, steps, warmup)
    # correctness of what was timed (outside the timed region)
    assert bool((res == 0).all()), "a chunk failed to decompress"
    assert torch.equal(out, data), "round trip is not byte-exact"
    assert bool((c_n > 0).all()) and bool((c_n <= comp_c.bound(fmt, size)).all())
    from tests import oracle_util
    ref = oracle_util.load_ref()
    nref = ref_verify(torch, ref, fmt, chunks, comp, c_off, c_n, first, len(chunks)) if ref else 0
    U = count * size
    C = int(c_n.sum().item())
    r = {"elapsed": elapsed, "t_comp": tm.span(0, 1), "t_dec": tm.span(1, 2),
         "U": U, "C": C, "verdict": verdict, "nref": nref,
         "tensors": (data, in_off, in_n, comp, c_off, c_av, c_n, out, res, comp_c, dec)}
    return r


def end_to_end(torch, dev, stream, api, tensors, slices=8):
    """The headline batch from pinned host memory and back, PCIe overlapped
    with compute (SURVEY.md 8(d) "(ii) end-to-end"; the reference times from
    host buffers, programs/test_util.c:143-164).  The batch is cut into
    `slices`: H2D of slice k + 1 (copy stream) runs beside the kernels and the
    device-side compaction of slice k (compute stream) and the D2H of slice
    k - 1 (second copy stream).  The host never waits in the middle of the
    pipeline for something that is not already done: the D2H of a slice is
    sized by its compacted total, read back by an async copy whose event has
    long fired when the host looks at it (the next slice is already
    enqueued).  Decompress direction alike: H2D of the compressed bytes of
    slice k + 1, inflate of slice k, D2H of the output of slice k - 1 - with
    one decompressor object and stream per slice, because a slice of streams
    is latency-bound (one wave per stream) and the slices run side by side."""
    data, in_off, in_n, comp, c_off, c_av, c_n, out, res, comp_c, dec = tensors
    n = in_off.numel()
    size = data.numel() // n
    bound = comp.numel() // n
    while n % slices:
        slices //= 2
    m = n // slices
    h_in = torch.empty(data.numel(), dtype=torch.uint8).pin_memory()
    h_in.copy_(data)
    h_comp = torch.empty(comp.numel(), dtype=torch.uint8).pin_memory()
    h_sizes = torch.empty(n, dtype=torch.int64).pin_memory()
    h_tot = torch.zeros(slices, dtype=torch.int64).pin_memory()
    h_out = torch.empty(out.numel(), dtype=torch.uint8).pin_memory()
    d_packed = torch.empty(comp.numel(), dtype=torch.uint8, device=dev)
    d_off = torch.empty(n, dtype=torch.int64, device=dev)
    d_sz = torch.empty(n, dtype=torch.int64, device=dev)
    s_in, s_out = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    s_dec = [torch.cuda.Stream(device=dev) for _ in range(slices)]
    decs = [dec] + [api.Decompressor() for _ in range(slices - 1)]
    rel_off = torch.arange(m, dtype=torch.int64, device=dev) * bound
    best_c = best_d = None
    total = 0
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        # ---- compress: host -> device -> host ----
        ev_in = [torch.cuda.Event() for _ in range(slices)]
        ev_k = [torch.cuda.Event() for _ in range(slices)]
        keep = []
        with torch.cuda.stream(s_in):
            for k in range(slices):
                data[k * m * size:(k + 1) * m * size].copy_(
                    h_in[k * m * size:(k + 1) * m * size], non_blocking=True)
                ev_in[k].record(s_in)
        at = 0

        def drain(j, at):
            ev_k[j].synchronize()               # fired long ago, except for the last slice
            tot = int(h_tot[j])
            with torch.cuda.stream(s_out):
                h_comp[at:at + tot].copy_(keep[j][0][:tot], non_blocking=True)
            return at + tot

        for k in range(slices):
            stream.wait_event(ev_in[k])
            sl = slice(k * m, (k + 1) * m)
            comp_c.compress_batch(FMT, data, in_off[sl], in_n[sl], comp, c_off[sl],
                                  c_av[sl], c_n[sl], stream=stream, max_chunk=size)
            packed, poff = api.compact_batch(comp[k * m * bound:(k + 1) * m * bound],
                                             rel_off, c_n[sl], stream=stream)
            keep.append((packed, poff))
            h_tot[k:k + 1].copy_(poff[m:m + 1], non_blocking=True)
            h_sizes[sl].copy_(c_n[sl], non_blocking=True)
            ev_k[k].record(stream)
            s_out.wait_event(ev_k[k])
            if k:
                at = drain(k - 1, at)
        at = drain(slices - 1, at)
        torch.cuda.synchronize()
        total = at
        t1 = time.perf_counter()
        # ---- decompress: host -> device -> host ----
        h_offs = torch.cumsum(h_sizes, 0) - h_sizes        # host-side index of the packed bytes
        ev_in = [torch.cuda.Event() for _ in range(slices)]
        ev_k = [torch.cuda.Event() for _ in range(slices)]
        with torch.cuda.stream(s_in):
            d_off.copy_(h_offs.pin_memory(), non_blocking=True)
            d_sz.copy_(h_sizes, non_blocking=True)
            for k in range(slices):
                lo = int(h_offs[k * m])
                hi = int(h_offs[(k + 1) * m]) if k + 1 < slices else total
                d_packed[lo:hi].copy_(h_comp[lo:hi], non_blocking=True)
                ev_in[k].record(s_in)
        for k in range(slices):
            # a slice of streams is latency-bound (one wave per stream), so the
            # slices run side by side: one decompressor object + stream each
            sk = s_dec[k]
            sk.wait_event(ev_in[k])
            sl = slice(k * m, (k + 1) * m)
            decs[k].decompress_batch(FMT, d_packed, d_off[sl], d_sz[sl], out, in_off[sl],
                                     in_n[sl], res[sl], stream=sk)
            with torch.cuda.stream(sk):
                h_out[k * m * size:(k + 1) * m * size].copy_(
                    out[k * m * size:(k + 1) * m * size], non_blocking=True)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if it:
            best_c = min(best_c or 1e9, t1 - t0)
            best_d = min(best_d or 1e9, t2 - t1)
    assert torch.equal(h_out, h_in) and bool((res == 0).all())
    U = data.numel()
    return {"compress_MBps": round(U / best_c / 1e6, 1),
            "decompress_MBps": round(U / best_d / 1e6, 1),
            "roundtrip_MBps": round(U / (best_c + best_d) / 1e6, 1),
            "compressed_bytes": total, "slices": slices,
            "note": "pinned host -> HBM -> pinned host, the batch in slices: H2D of "
                    "slice k + 1 beside the kernels (+ device compaction) of slice k "
                    "beside the D2H of slice k - 1, three streams, no host wait inside "
                    "the pipeline; best of 2 after a warm-up; one GPU"}


def extra_roundtrip(name, workload, torch, dist, world, rank, dev, stream, api,
                    shard, chunks, total, fmt, level, steps, scaling, cpu=True,
                    pmc_of=None):
    """A non-headline BASELINE config: compress + decompress of `total`
    chunks (strong: partitioned over the ranks; weak: `total` per rank)."""
    if scaling == "strong":
        lo, hi = shard.partition(total, world, rank)
    else:
        lo, hi = rank * total, (rank + 1) * total
    r = roundtrip_config(torch, dist, world, rank, dev, stream, api, shard,
                         chunks, hi - lo, lo, fmt, level, steps, 1)
    U, C, tc, td = all_sum(torch, dist, dev, r["U"], r["C"], r["t_comp"], r["t_dec"])
    tc, td = tc / world, td / world     # mean launch time over ranks
    ms = r["elapsed"] / steps * 1e3
    del r["tensors"]
    torch.cuda.empty_cache()
    return {"cpu_baseline": cpu_baseline(chunks, fmt, level) if cpu and rank == 0 else None,
            "workload": workload, "scaling": scaling, "n_gpus": world,
            "chunks_total": int(round(U / len(chunks[0]))),
            "verdicts": {"chunks": int(r["verdict"][0]), "failed": int(r["verdict"][1])},
            "steps": steps, "ms_per_step": round(ms, 3),
            "roundtrip_MBps": round(U / (ms / 1e3) / 1e6, 1),
            "compress_MBps": round(U / tc / 1e6, 1),
            "decompress_MBps": round(U / td / 1e6, 1),
            "compress_ms": round(tc * 1e3, 3), "decompress_ms": round(td * 1e3, 3),
            "compressed_ratio": round(C / U, 4),
            "verified": "round trip byte-exact on every rank; " +
                        (f"every distinct compressed stream ({r['nref']} per rank) decoded by "
                         "the real reference (oracle/_ref) to the original bytes"
                         if r["nref"] else "oracle/_ref not on this box"),
            "roofline": {"bound": "valu-issue", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                         "note": "frac = algorithmic bytes / launch time / HBM peak (the "
                                 "contract's figure); what binds is VALU issue: issue_frac",
                         "compress": dict(
                             {"achieved": round((U + C) / world / tc / 1e9, 2),
                              "frac": round((U + C) / world / tc / 1e9 / HBM_PEAK_GBS, 5)},
                             **({"issue": issue_roof(*pmc_of[0], tc, pmc_of[1])}
                                if pmc_of else {})),
                         "decompress": {"achieved": round((U + C) / world / td / 1e9, 2),
                                        "frac": round((U + C) / world / td / 1e9 / HBM_PEAK_GBS, 5)}}}


def extra_inflate(torch, dist, world, rank, dev, stream, api, shard, total, steps, cpu=True):
    """BASELINE configs[3]: `total` gzip streams pre-compressed by the
    reference (oracle/_ref) at level 6 OUTSIDE the timed region, contiguous
    shard per rank, decompress-only in exact-fill mode
    (actual_out_nbytes_ret = NULL, out_avail = 65 536), every byte of the
    output compared, verdicts gathered to rank 0."""
    from tests import datagen, oracle_util
    ref = oracle_util.load_ref()
    if ref is None:
        return {"skipped": "oracle/_ref/libdeflate_ref.so not built"}
    distinct = min(4096, total)
    chunks = gen_chunks(distinct, CHUNK, 0x0E110004)
    streams = [ref.compress("gzip", 6, c) for c in chunks]
    lo, hi = shard.partition(total, world, rank)
    n = hi - lo
    # one aligned blob of the distinct streams, repeated on the device
    offs, blob = [], bytearray()
    for s in streams:
        offs.append(len(blob))
        blob += s
        blob += bytes((-len(blob)) % 16)
    blen = len(blob)
    reps = (n + distinct - 1) // distinct + 1
    d_blob = torch.frombuffer(blob, dtype=torch.uint8).to(dev).repeat(reps)
    d_in = torch.cat([d_blob, torch.zeros(64, dtype=torch.uint8, device=dev)])
    gi = torch.arange(lo, hi, device=dev)
    first_rep = lo // distinct
    base_off = torch.tensor(offs, dtype=torch.int64, device=dev)
    in_off = base_off[gi % distinct] + (gi // distinct - first_rep) * blen
    in_n = torch.tensor([len(s) for s in streams], dtype=torch.int64,
                        device=dev)[gi % distinct]
    want_base = torch.frombuffer(bytearray(b"".join(chunks)), dtype=torch.uint8).to(dev)
    out = torch.zeros(n * CHUNK, dtype=torch.uint8, device=dev)
    out_off = torch.arange(n, dtype=torch.int64, device=dev) * CHUNK
    out_av = torch.full((n,), CHUNK, dtype=torch.int64, device=dev)
    res = torch.full((n,), -1, dtype=torch.int32, device=dev)
    dec = api.Decompressor()
    tm = Timer(torch, dist, dev, stream, steps, 2)

    def step(ev):
        if ev:
            ev[0].record(stream)
        dec.decompress_batch("gzip", d_in, in_off, in_n, out, out_off, out_av,
                             res, stream=stream)   # actual_out = None: exact fill
        if ev:
            ev[1].record(stream)
        return shard.gather_verdicts(in_n, res, dist, world)

    elapsed, verdict = tm.run(step, steps, 1)
    assert bool((res == 0).all()), "a stream failed to decompress"
    want = want_base.view(distinct, CHUNK)[gi % distinct].reshape(-1)
    assert torch.equal(out, want), "decompressed bytes differ from the original"
    U, C, td = all_sum(torch, dist, dev, n * CHUNK, int(in_n.sum().item()), tm.span(0, 1))
    td /= world
    ms = elapsed / steps * 1e3
    r = {"cpu_baseline": cpu_baseline(chunks, "gzip", 6, "d") if cpu and rank == 0 else None,
         "workload": f"configs[3]: decompress-only, {total} gzip streams of 64 KiB "
                     f"chunks ({distinct} distinct) compressed by the reference at level 6, "
                     "contiguous shard per rank",
         "scaling": "strong", "n_gpus": world, "steps": steps,
         "streams_total": int(round(U / CHUNK)),
         "verdicts": {"chunks": int(verdict[0]), "failed": int(verdict[1])},
         "ms_per_step": round(ms, 3),
         "decompress_MBps": round(U / (ms / 1e3) / 1e6, 1),
         "kernel_ms": round(td * 1e3, 3),
         "compressed_ratio": round(C / U, 4),
         "verified": f"all {int(U)} output bytes equal the original; "
                     f"{verdict[0]} verdicts gathered, {verdict[1]} failed",
         "roofline": {"bound": "valu-issue", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                      "achieved": round((U + C) / world / td / 1e9, 2),
                      "frac": round((U + C) / world / td / 1e9 / HBM_PEAK_GBS, 5),
                      "issue": issue_roof("inflate64k", "lda_inflate_wave_kernel", td,
                                          (U / world / CHUNK) / 65536.0)}}
    del d_in, d_blob, out, want
    torch.cuda.empty_cache()
    return r


def single_stream(mib=16, cpu=True):
    """ONE large buffer through the reference's own
    single-buffer calls (host pointers in and out, the shape of
    programs/gzip.c:149-303): libdeflate_gzip_decompress on a stream the
    REFERENCE compressed at level 6, libdeflate_gzip_compress on the same
    text; best of 5 calls after a warm-up, buffers reused like
    programs/benchmark.c does.  Beside it the reference itself on one host
    core (one call is one thread there).  Host to host: PCIe and the staging
    copies are inside the timed calls."""
    import ctypes
    import numpy as np
    from libdeflate_amd import api, binding
    from tests import datagen, oracle_util
    ref = oracle_util.load_ref()
    if ref is None:
        return {"skipped": "oracle/_ref/libdeflate_ref.so not built"}
    n = mib << 20
    data = datagen.text_chunk(n, 0x0E110006)
    zref = np.frombuffer(ref.compress("gzip", LEVEL, data), dtype=np.uint8)
    src = np.frombuffer(data, dtype=np.uint8)
    back = np.zeros(n, dtype=np.uint8)
    c, d = api.Compressor(LEVEL), api.Decompressor()
    lib = binding.load()
    bound = c.bound("gzip", n)
    zout = np.zeros(bound, dtype=np.uint8)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    ao = ctypes.c_size_t(0)
    t_d = t_c = 1e9
    zn = 0
    for it in range(6):
        t0 = time.perf_counter()
        r = lib.libdeflate_gzip_decompress(d._h, P(zref), zref.size, P(back), n, ctypes.byref(ao))
        t1 = time.perf_counter()
        zn = lib.libdeflate_gzip_compress(c._h, P(src), n, P(zout), bound)
        t2 = time.perf_counter()
        assert r == 0 and ao.value == n and zn
        if it:
            t_d, t_c = min(t_d, t1 - t0), min(t_c, t2 - t1)
    st = None
    lib.libdeflate_gzip_decompress(d._h, P(zref), zref.size, P(back), n, ctypes.byref(ao))
    st = binding.stream_stats()
    assert back.tobytes() == data, "single-stream decompress differs"
    rr = ref.decompress_ex("gzip", zout[:zn].tobytes(), n)
    assert rr[0] == 0 and rr[3] == data, "the reference rejects the segmented stream"
    cpu = cpu_baseline([data], "gzip", LEVEL, only_t1=True) if cpu else None
    out = {"workload": f"one {mib} MiB enwik-style buffer, gzip level {LEVEL}: "
                       "libdeflate_gzip_decompress of the reference's stream and "
                       "libdeflate_gzip_compress, host pointers in and out, one call each",
           "decompress_MBps": round(n / t_d / 1e6, 1), "decompress_ms": round(t_d * 1e3, 3),
           "compress_MBps": round(n / t_c / 1e6, 1), "compress_ms": round(t_c * 1e3, 3),
           "compressed_ratio": round(zn / n, 4),
           "reference_ratio": round(zref.size / n, 4),
           "decompress_path": {k: st[k] for k in ("parallel", "blocks_found", "chunks_decoded",
                                                    "repairs", "us_in", "us_find", "us_count",
                                                    "us_decode", "us_sum", "us_out")},
           "verified": "decompressed bytes equal the original; the compressed stream "
                       "decoded by the real reference to the original",
           "cpu_baseline": None}
    if cpu and "t1" in cpu:
        out["cpu_baseline"] = {
            "kind": "reference", "cores": 1, "unit": "MB/s",
            "decompress_MBps": cpu["t1"]["decompress_MBps"],
            "compress_MBps": cpu["t1"]["compress_MBps"],
            "sample": f"the same {mib} MiB buffer, one thread (a single-buffer call of the "
                      "reference runs on one core), oracle/cpu_bench.c, best of 3",
            "cpu_model": cpu["cpu_model"]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chunks", type=int, default=CHUNKS_PER_GPU)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--configs", default="all", choices=["all", "headline"],
                    help="'headline' skips the other BASELINE configs")
    ap.add_argument("--streams", type=int, default=65536,
                    help="configs[3] stream count (total over all ranks)")
    ap.add_argument("--blocks", type=int, default=1 << 20,
                    help="configs[4] 4 KiB block count (total over all ranks)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the verdict gather: nccl "
                         "(= RCCL over xGMI, the default) or gloo (collectives on "
                         "host tensors: lets N ranks share one GPU in the test "
                         "of the N > 1 control flow)")
    ap.add_argument("--one-device", action="store_true",
                    help="every rank uses cuda:0 (with --backend gloo: the "
                         "multi-rank path on a single-GPU box)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run every collective of "
                         "the N > 1 path even at world size 1 (RCCL executed on a "
                         "single-GPU box: tests/test_rccl_gpu.py)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch(a))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.one_device:
        assert a.backend == "gloo" or world == 1, "RCCL needs one GPU per rank"
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or a.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if a.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from libdeflate_amd import api, shard
    from tests import datagen
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.current_stream()
    n = a.chunks
    # 64 KiB chunk mix (5 text, binary, low-entropy,
    # random per 8), every chunk of a rank's batch with its own seed
    chunks = gen_chunks(n, CHUNK, 0x0E110003 + rank * 100003)

    head = roundtrip_config(torch, dist, world, rank, dev, stream, api, shard,
                            chunks, n, 0, FMT, LEVEL, a.steps, a.warmup)
    elapsed, t_comp, t_dec = head["elapsed"], head["t_comp"], head["t_dec"]
    U, C = head["U"], head["C"]
    e2e = None
    if rank == 0 and a.configs == "all":
        e2e = end_to_end(torch, dev, stream, api, head["tensors"])
    del head["tensors"]
    torch.cuda.empty_cache()

    extras = {}
    if a.configs == "all":
        extras["configs[1]"] = extra_roundtrip(
            "l1", "configs[1]: 4096 x 64 KiB raw DEFLATE buffers, level 1, "
            "compress then decompress, per GPU", torch, dist, world, rank, dev,
            stream, api, shard,
            gen_chunks(a.chunks, CHUNK, 0x0E110002 + rank * 100003),
            a.chunks, "deflate", 1, 3, "weak", cpu=not a.no_cpu,
            pmc_of=(("l1", "lda_deflate_batch_kernel"), a.chunks / 4096.0))
        extras["configs[3]"] = extra_inflate(torch, dist, world, rank, dev, stream,
                                             api, shard, a.streams, 3, cpu=not a.no_cpu)
        extras["configs[4]"] = extra_roundtrip(
            "zlib4k", f"configs[4]: {a.blocks} x 4 KiB zlib buffers "
            "(filesystem-block mix), level 9, compress then decompress, "
            "partitioned over the GPUs", torch, dist, world, rank, dev, stream,
            api, shard,
            gen_chunks(min(a.blocks, 16384), 4096, 0x0E110005, mix4k=True),
            a.blocks, "zlib", 9, 2, "strong", cpu=not a.no_cpu,
            pmc_of=(("small", "lda_deflate_small_kernel"), a.blocks / world / 262144.0))
        if rank == 0:
            extras["single_stream"] = single_stream(cpu=not a.no_cpu)

    if rank == 0:
        total_chunks, n_fail = head["verdict"]
        ms = elapsed / a.steps * 1e3
        value = U * world / (elapsed / a.steps) / 1e6
        traffic, valu, pmc_src = pmc_static()
        line = {
            "metric": "compress+decompress MB/s, 64 KiB chunks level 6",
            "value": round(value, 1), "unit": "MB/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic (seeded 64 KiB chunk mix: 5 text, 1 binary, "
                    f"1 low-entropy, 1 random per 8; {n} distinct chunks per GPU, "
                    "chunk i seeded base + i)",
            "config": {"workload": "configs[2]: 4096 x 64 KiB gzip buffers, "
                       "level 6 + CRC-32, compress then decompress, per GPU",
                       "chunks_per_gpu": n, "chunk_bytes": CHUNK,
                       "format": FMT, "level": LEVEL,
                       "parallelism": f"shard{world} (independent chunks; "
                       "RCCL gather of verdicts only)"},
            "compress_MBps": round(U * world / t_comp / 1e6, 1),
            "decompress_MBps": round(U * world / t_dec / 1e6, 1),
            "compressed_ratio": round(C / U, 4),
            "verdicts": {"chunks": int(total_chunks), "failed": int(n_fail)},
            "collectives": {"backend": dist.get_backend() if dist else None, "world": world,
                            "forced_at_world_1": bool(a.force_dist and world == 1)},
            "verified": "round trip byte-exact on every rank (torch.equal over the batch); " +
                        (f"all {head['nref']} compressed streams of rank 0 decoded by the real "
                         "reference (oracle/_ref) to the original bytes, outside the timed region"
                         if head["nref"] else "oracle/_ref not on this box"),
            "roofline": {
                "bound": "valu-issue", "kernel": "lda_deflate_batch_kernel",
                "bound_note": "frac (below) = algorithmic bytes / launch time / HBM peak, "
                              "the contract's figure; the binding resource is VALU issue "
                              "and the latency of the waves' dependent LDS chains: issue_frac",
                "issue_frac": (issue_roof("bench", "lda_deflate_batch_kernel", t_comp) or
                               {}).get("issue_frac"),
                "achieved": round((U + C) / t_comp / 1e9, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round((U + C) / t_comp / 1e9 / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                "traffic_source": f"{pmc_src} (static: PMC passes of "
                                  "tools/prof_pmc.sh, not this run)" if pmc_src else None,
                "issue": None if not valu else {
                    "valu_wave_insts_per_launch": int(valu),
                    "simd_issue_frac": round(valu * 4 / (1024 * 2.4e9 * t_comp), 3),
                    "source": f"{pmc_src} (static) / this run's launch time",
                    "note": "VALU wave-instructions x 4 cycles / (1024 SIMDs x "
                            "2.4 GHz x launch time): the kernel is issue-bound, "
                            "not HBM-bound"},
                "algorithmic_bytes_per_launch": U + C,
                "avg_launch_ms": round(t_comp * 1e3, 3),
                "note": "HIP events on the launch stream around the compress "
                        "call; U + C with the CRC-32 pass counted as part of it",
                "inflate_kernel": dict({
                    "kernel": "lda_inflate_wave_kernel",
                    "achieved": round((U + C) / t_dec / 1e9, 2),
                    "frac": round((U + C) / t_dec / 1e9 / HBM_PEAK_GBS, 5),
                    "avg_launch_ms": round(t_dec * 1e3, 3)}, **inflate_static(t_dec)),
            },
        }
        if e2e:
            line["end_to_end"] = e2e
        if extras:
            line["configs"] = extras
        line["cpu_baseline"] = None if a.no_cpu else cpu_baseline(chunks, FMT, LEVEL)
        print(json.dumps(line))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
