"""Round 6: does a high-priority stream with a PENDING WAIT at its head slow the launches of a normal-priority stream?  (hypothesis: only when their
hardware queues share a CP pipe.)  Streams first used in creation order n0.., h0.. as in r06_stream_sets.py, where (tok=h1|h2; dense=n0,n1) was 13 ms and
(tok=h0|h3|h4) 7.2 ms."""
import time
import torch

torch.cuda.init()
N, NH = 8, 5
ns = [torch.cuda.Stream() for _ in range(N)]
hs = [torch.cuda.Stream(priority=-1) for _ in range(NH)]
x = torch.zeros(8, device="cuda")
for s in ns + hs:
    with torch.cuda.stream(s):
        x.add_(1)
torch.cuda.synchronize()
null = torch.cuda.default_stream()


def chain_graph(d, n=200):
    y = torch.zeros(64, device="cuda")
    with torch.cuda.stream(d):
        for _ in range(3):
            y.add_(1)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=d):
        for _ in range(n):
            y.add_(1)
    torch.cuda.synchronize()
    return g, y


def chain_ms(d, g, h=None, block_cycles=8000000):
    torch.cuda.synchronize()
    if h is not None:
        torch.cuda._sleep(block_cycles)          # on the default stream: ~3.4 ms
        ev = torch.cuda.Event(); ev.record(null)
        h.wait_event(ev)                         # h's queue now holds an unsatisfied barrier
        with torch.cuda.stream(h):
            x.add_(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(d):
        e0.record(d); g.replay(); e1.record(d)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for di in (0, 1, 2):
    g, y = chain_graph(ns[di])
    free = min(chain_ms(ns[di], g) for _ in range(3))
    row = [f"n{di}: free {free:.3f} ms |"]
    for hi in range(NH):
        t = min(chain_ms(ns[di], g, hs[hi]) for _ in range(3))
        row.append(f"h{hi} blocked: {t:.3f}")
    print(" ".join(row), flush=True)
