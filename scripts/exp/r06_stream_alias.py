"""Which pairs of torch pool streams run concurrently?  One spinning kernel (torch.cuda._sleep, one workgroup) on each of two streams: ~1x the
single time = concurrent (different HSA queues), ~2x = serialised (the two streams share a hardware queue, or sit behind each other's barriers).
Streams: 'null', normal-priority pool streams n0.. (in the order torch.cuda.Stream() hands them out), high-priority h0.."""
import os
import sys
import time

import torch

torch.cuda.init()
x = torch.zeros(1, device="cuda")
names, streams = ["null"], [torch.cuda.default_stream()]
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    names.append(f"n{i}"); streams.append(torch.cuda.Stream())
for i in range(4):
    names.append(f"h{i}"); streams.append(torch.cuda.Stream(priority=-1))
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"), " stream ids:", {n: s.stream_id for n, s in zip(names, streams)})


def t_of(pairs, cycles):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in pairs:
        with torch.cuda.stream(s):
            torch.cuda._sleep(cycles)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


cycles = 100000
t_of([streams[1]], cycles)
while t_of([streams[1]], cycles) < 1.0:
    cycles *= 2
single = min(t_of([streams[1]], cycles) for _ in range(3))
print(f"single sleep({cycles}) = {single:.2f} ms")
print("      " + " ".join(f"{n:>5}" for n in names))
for i, a in enumerate(names):
    row = []
    for j, b in enumerate(names):
        if j <= i:
            row.append("    .")
            continue
        t = min(t_of([streams[i], streams[j]], cycles) for _ in range(2))
        row.append(f"{t / single:5.2f}")
    print(f"{a:>5} " + " ".join(row))
