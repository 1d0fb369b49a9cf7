"""d2h_probe.py - device -> pinned host copies of 16 MiB: one copy, four slices on one
stream, four slices on two / four streams (is one copy kernel the link's speed?).  A probe."""
import time
import torch

n = 16 << 20
d = torch.empty(n, dtype=torch.uint8, device="cuda")
h = torch.empty(n, dtype=torch.uint8).pin_memory()
streams = [torch.cuda.Stream() for _ in range(4)]


def run(nslices, nstreams, reps=20):
    sl = n // nslices
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(nslices):
            with torch.cuda.stream(streams[i % nstreams]):
                h[i * sl:(i + 1) * sl].copy_(d[i * sl:(i + 1) * sl], non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


for ns, nst in ((1, 1), (4, 1), (4, 2), (4, 4), (8, 1), (8, 2), (16, 4)):
    t = run(ns, nst)
    print(f"{ns} slices on {nst} streams: {t * 1e6:.0f} us = {n / t / 1e9:.1f} GB/s")
# host -> device for comparison
for ns, nst in ((1, 1), (4, 2)):
    sl = n // ns
    best = 1e9
    for _ in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(ns):
            with torch.cuda.stream(streams[i % nst]):
                d[i * sl:(i + 1) * sl].copy_(h[i * sl:(i + 1) * sl], non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"H2D {ns} slices on {nst} streams: {best * 1e6:.0f} us = {n / best / 1e9:.1f} GB/s")
