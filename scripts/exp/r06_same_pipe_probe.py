"""Round 6: two NORMAL-priority hardware queues on one command-processor pipe.  First-use order n0, n1, h0, h1, n2 makes n2's queue the 6th of the process
(same pipe as n0's, the 2nd, if pipes are dealt round-robin over four).  Probes: k back-to-back spinning kernels on each of two streams (k = 1, 2, 3):
concurrent = k x single, serialised = 2k x single; and a 200-launch chain of tiny kernels on each."""
import time
import torch

torch.cuda.init()
x = torch.zeros(8, device="cuda")
null = torch.cuda.default_stream()
def use(s):
    with torch.cuda.stream(s):
        x.add_(1)
    torch.cuda.synchronize()
    return s
n0 = use(torch.cuda.Stream()); n1 = use(torch.cuda.Stream())
h0 = use(torch.cuda.Stream(priority=-1)); h1 = use(torch.cuda.Stream(priority=-1))
n2 = use(torch.cuda.Stream())
n3 = use(torch.cuda.Stream()); n4 = use(torch.cuda.Stream()); n5 = use(torch.cuda.Stream())
S = {"null": null, "n0": n0, "n1": n1, "n2": n2, "n3": n3, "n4": n4, "n5": n5}
cycles = 1000000


def t_of(streams, k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in streams:
        with torch.cuda.stream(s):
            for _ in range(k):
                torch.cuda._sleep(cycles)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


single = min(t_of([n0], 1) for _ in range(3))
print(f"single = {single:.3f} ms")
names = list(S)
for k in (1, 2, 3):
    print(f"k = {k} sleeps per stream: time / (k x single)")
    for i, a in enumerate(names):
        row = []
        for j, b in enumerate(names):
            if j <= i:
                row.append("    ."); continue
            row.append(f"{min(t_of([S[a], S[b]], k) for _ in range(2)) / (k * single):5.2f}")
        print(f"{a:>5} " + " ".join(row))


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


graphs = {k: chain_graph(v) for k, v in S.items() if k != "null"}
def both(a, b):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in (a, b) if b else (a,):
        with torch.cuda.stream(S[k]):
            graphs[k][0].replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
print("200-launch chains: alone", {k: round(min(both(k, None) for _ in range(3)), 3) for k in graphs})
for a in ("n0",):
    print("pairs with n0:", {b: round(min(both(a, b) for _ in range(3)), 3) for b in graphs if b != a})
print("pairs with n1:", {b: round(min(both("n1", b) for _ in range(3)), 3) for b in graphs if b != "n1"})


def chain_while_blocked(d, b, cycles=6000000):
    """200-launch chain on d while NORMAL-priority stream b holds an unsatisfied wait (the default stream sleeps)."""
    torch.cuda.synchronize()
    torch.cuda._sleep(cycles)
    ev = torch.cuda.Event(); ev.record(null)
    S[b].wait_event(ev)
    with torch.cuda.stream(S[b]):
        x.add_(1)
    t0 = time.perf_counter()
    with torch.cuda.stream(S[d]):
        graphs[d][0].replay()
    S[d].synchronize()
    t = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    return t
print("chain on n0 while a normal-priority stream waits (sleep = %.2f ms):" % (min(t_of([null], 1) for _ in range(2)) * 6), {b: round(min(chain_while_blocked("n0", b) for _ in range(3)), 3) for b in ("n1", "n2", "n3", "n4")})
print("chain on n1 while a normal-priority stream waits:", {b: round(min(chain_while_blocked("n1", b) for _ in range(3)), 3) for b in ("n0", "n2", "n3", "n5")})
