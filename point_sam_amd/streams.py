"""The process-wide HIP streams of the pipelines, chosen by PROBING how the runtime mapped them onto the hardware.

What round 6 found (profiles/r06/r06_inproc.txt, r06_stream_alias.txt, r06_pipe_probe.txt, r06_same_pipe_probe.txt; scripts/exp/r06_*.py):

* ROCm (ROCclr, 7.x) folds the HIP streams of a process onto at most GPU_MAX_HW_QUEUES (default 4) hardware queues per priority, in order of FIRST USE,
  and the hardware queues onto the four pipes of the compute command processor, in order of creation.  torch.cuda.Stream() hands out the streams of a
  32-entry pool round-robin and never takes one back.
* Two streams on one hardware queue run in order; a `wait_event` queued on one holds back the other (a dense stream that shares the DEFAULT stream's
  queue sits behind the caller's wait for the previous batch).
* Two normal-priority queues on one PIPE that both launch kernels back to back pay ~3 us per launch for the pipe's queue switches (a chain of 200 tiny
  launches: 0.33 ms alone, 0.46 ms beside a chain on another pipe, 0.88 ms beside one on the same pipe -- slower than sharing the queue, 0.65 ms).
* A normal-priority queue that shares a pipe with a HIGH-priority queue makes NO progress while the high-priority queue has an unsatisfied barrier
  packet at its head (the 200-launch chain takes the whole 3.4 ms of the wait; exactly one of the four high-priority queues does this to a given
  normal queue).  In a pipeline the tokenizer stream of a later batch waits for an earlier batch's dense stage: a dense stream starved by it runs
  slower than with no overlap at all.

The streams the FIRST pipeline of a fresh process got happened to sit on four different pipes.  A second pipeline's did not: 7.2 -> 13.6 ms per ViT-g
session, 9.7 -> 12.6 ms per cfg #3 cloud, with every graph on its own exactly as fast as before, the first pipeline still fast when re-run, teardown of
the first pipeline no cure, GPU_MAX_HW_QUEUES=8 no cure.  Hence: ONE set of streams per process and device, shared by every pipeline (streams are
in-order, so sharing is ordered, never wrong), each stream admitted only after the probes below; candidates that fail are passed over.
"""
import os
import threading
import time

import torch

_POOL = {}
_LOCK = threading.Lock()      # pipelines may be built from several host threads (a server): the probes must not interleave


def _timed_on(streams, fn, device):
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for s in streams:
        with torch.cuda.stream(s):
            fn(s)
    for s in streams:
        s.synchronize()
    return time.perf_counter() - t0


def streams_alias(a, b, cycles: int = 400000) -> bool:
    """True if streams `a` and `b` share a hardware queue: one spinning workgroup (`torch.cuda._sleep`) on each takes 2x the single time instead of 1x."""
    spin = lambda s: torch.cuda._sleep(cycles)
    _timed_on((a,), spin, a.device); _timed_on((b,), spin, a.device)      # first use maps a stream to its hardware queue
    single = min(_timed_on((a,), spin, a.device) for _ in range(2))
    return min(_timed_on((a, b), spin, a.device) for _ in range(2)) > 1.5 * single


class _Chain:
    """`n` dependent tiny launches on stream `s`, captured once: replayed, its duration is the stream's launch rate (the command processor's, not Python's)."""

    def __init__(self, s, n: int = 150, capture_on=None):
        self.y = torch.zeros(64, device=s.device)
        s = capture_on if capture_on is not None else s      # (a graph cannot be captured on the default stream; it can be replayed there)
        with torch.cuda.stream(s):
            for _ in range(2):
                self.y.add_(1)
        torch.cuda.synchronize(s.device)
        self.g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g, stream=s):
            for _ in range(n):
                self.y.add_(1)
        torch.cuda.synchronize(s.device)


def launch_chains_conflict(a, b) -> bool:
    """True if back-to-back launches on `a` and on `b` hold each other up: same hardware queue (~1.9x the time of one chain) or two queues on one
    command-processor pipe (~2.5x); queues on different pipes: ~1.35x."""
    ca, cb = _Chain(a), _Chain(b)
    by = {id(a): ca, id(b): cb}
    run = lambda s: by[id(s)].g.replay()
    alone = max(min(_timed_on((s,), run, a.device) for _ in range(3)) for s in (a, b))
    return min(_timed_on((a, b), run, a.device) for _ in range(3)) > 1.6 * alone


def _chain_while_waiting(d, waiter, sleeper, cycles):
    """(free, blocked, wait) seconds: a launch chain on `d` alone, the same chain while stream `waiter` holds an unsatisfied wait at its head (for a
    spin of `cycles` on stream `sleeper`), and the spin itself."""
    dev = d.device
    c = _Chain(d, capture_on=sleeper if d.cuda_stream == torch.cuda.default_stream(dev).cuda_stream else None)
    z = torch.zeros(64, device=dev)

    def go(block):
        torch.cuda.synchronize(dev)
        if block:
            with torch.cuda.stream(sleeper):
                torch.cuda._sleep(cycles)
                ev = torch.cuda.Event()
                ev.record(sleeper)
            waiter.wait_event(ev)
            with torch.cuda.stream(waiter):
                z.add_(1)
        t0 = time.perf_counter()      # no device synchronisation here: the wait must still be pending while the chain runs
        with torch.cuda.stream(d):
            c.g.replay()
        d.synchronize()
        t = time.perf_counter() - t0
        torch.cuda.synchronize(dev)
        return t

    go(False)
    free = min(go(False) for _ in range(3))
    wait = min(_timed_on((sleeper,), lambda s: torch.cuda._sleep(cycles), dev) for _ in range(2))
    return free, min(go(True) for _ in range(3)), wait


def stream_starved_by(d, h, sleeper=None, cycles: int = 3000000) -> bool:
    """True if stream `d` makes no progress while high-priority stream `h` has an unsatisfied wait at its head (their hardware queues share a
    command-processor pipe): a chain of queued launches on `d` then takes the whole wait instead of its own ~0.3 ms.  (Launches that arrive one by one
    at an empty queue are NOT held back -- the probe queues its chain at once, as a graph replay does.)  The blocker spins on `sleeper` (default: the
    default stream), which must not share a queue with `d`."""
    free, blocked, wait = _chain_while_waiting(d, h, sleeper if sleeper is not None else torch.cuda.default_stream(d.device), cycles)
    return blocked > free + 0.5 * wait


def mini_pipeline_ms(tok, dense, steps: int = 12, slots: int = 4, launches: int = 120, cycles: int = 20000, tok_cycles: int = 600000):
    """A pipeline in miniature on the given streams, with the event structure of GraphPipeline.submit / next (inputs touched on the caller's stream; the
    tokenizer graph of a slot waits for the caller's stream, which waits for the slot's previous dense graph; the dense graph waits for the tokenizer
    graph) and one-workgroup spinning kernels in place of the real ones: `launches` x `cycles` per dense graph, one `tok_cycles` spin per tokenizer graph.
    Returns (ms per step, ms of one dense graph alone): with len(dense) streams that overlap the ratio is ~1 / len(dense); on a bad hardware-queue
    mapping it is ~1 or worse."""
    dev = tok.device
    main = torch.cuda.current_stream(dev)
    x = torch.zeros(64, device=dev)
    S = []
    for i in range(slots):
        ds = dense[i % len(dense)]
        for s_, n_, c_ in ((tok, 1, tok_cycles), (ds, 2, cycles)):
            with torch.cuda.stream(s_):
                torch.cuda._sleep(c_)
        torch.cuda.synchronize(dev)
        gt, gd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(gt, stream=tok):
            torch.cuda._sleep(tok_cycles)
        with torch.cuda.graph(gd, stream=ds):
            for _ in range(launches):
                torch.cuda._sleep(cycles)
        S.append((gt, gd, ds, torch.cuda.Event(), torch.cuda.Event()))
    torch.cuda.synchronize(dev)

    def submit(i):
        gt, gd, ds, tok_done, done = S[i % slots]
        x.add_(1)
        tok.wait_stream(main)
        with torch.cuda.stream(tok):
            gt.replay()
            tok_done.record(tok)
        ds.wait_stream(main)
        ds.wait_event(tok_done)
        with torch.cuda.stream(ds):
            gd.replay()
            done.record(ds)

    def run(n):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(min(slots, n)):
            submit(i)
        for k in range(n):
            main.wait_event(S[k % slots][4])
            if k + slots < n:
                submit(k + slots)
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / n * 1e3

    run(slots)
    per_step = min(run(steps) for _ in range(2))
    gd, ds = S[0][1], S[0][2]
    alone = min(_timed_on((ds,), lambda s: gd.replay(), dev) for _ in range(2)) * 1e3
    return per_step, alone


def pipeline_streams(device, dense: int):
    """(tokenizer stream, [dense streams]) of the process for `device`: one high-priority stream for the coordinate-only tokenizer stage and the first
    `dense` dense-stage streams, created on first request, shared by every BatchPipeline / GraphPipeline.

    Admission (at least two dense streams are always probed, so the tokenizer stream is validated even for a pipeline without dense streams): a dense
    candidate must not share a hardware queue with the default stream or the caller's current stream (`streams_alias`), must not be starved by the
    tokenizer candidate (`stream_starved_by`) and must not conflict with the dense streams already admitted (`launch_chains_conflict`); the tokenizer
    candidate must not starve the default stream either; finally the set must run the miniature pipeline (`mini_pipeline_ms`) with its dense graphs
    overlapping (ms per step < 0.8 x one dense graph).  Up to 6 tokenizer candidates x 10 dense candidates; if no set passes (GPU_MAX_HW_QUEUES=2, say)
    the one with the best miniature-pipeline ratio is kept and the pool is marked `compromised` (`pipeline_streams_report`).  More than two dense
    streams: a third is admitted by the same probes if the hardware queues allow it, otherwise taken unprobed (counted, not `compromised`).
    PSAM_PRIVATE_STREAMS=1: every call returns fresh, unprobed streams (the behaviour until round 5, kept for the A/B)."""
    if os.environ.get("PSAM_PRIVATE_STREAMS", "0") == "1" or not hasattr(torch.cuda, "_sleep"):      # (no spin kernel to probe with: unprobed streams)
        return torch.cuda.Stream(device=device, priority=-1), [torch.cuda.Stream(device=device) for _ in range(dense)]
    with _LOCK:
        return _pipeline_streams_locked(device, dense)


def _pipeline_streams_locked(device, dense: int):
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    ent = _POOL.setdefault((dev.type, idx), {"tok": None, "dense": [], "cands": [], "passed_over": 0, "compromised": False, "unadmitted": 0, "probe_s": 0.0, "mini_ratio": None})
    want = max(2, dense)
    if ent["tok"] is not None and len(ent["dense"]) >= want:
        return ent["tok"], list(ent["dense"][:dense])
    t0 = time.perf_counter()
    with torch.cuda.device(idx):
        null = torch.cuda.default_stream(idx)
        cur = torch.cuda.current_stream(idx)
        mains = [null] + ([cur] if cur.cuda_stream != null.cuda_stream else [])
        memo = {}

        def cached(kind, fn, a, b):
            key = (kind, a.cuda_stream, b.cuda_stream)
            if key not in memo:
                memo[key] = fn(a, b)
            return memo[key]

        def admit(tok, have):
            got, k = list(have), 0
            while len(got) < want and k < 10:
                if k == len(ent["cands"]):
                    ent["cands"].append(torch.cuda.Stream(device=device))
                c = ent["cands"][k]
                k += 1
                if any(c.cuda_stream == g.cuda_stream for g in got) or c.cuda_stream == tok.cuda_stream:
                    continue
                if (not any(cached("alias", streams_alias, c, m) for m in mains)
                        and not cached("starve", stream_starved_by, c, tok)
                        and not any(cached("chains", launch_chains_conflict, c, g) for g in got)):
                    got.append(c)
            return got

        def ratio(tok, got):
            per, alone = mini_pipeline_ms(tok, got)
            return per / alone

        if ent["tok"] is not None:      # a later request for more dense streams: the tokenizer stream stays
            ent["dense"] = admit(ent["tok"], ent["dense"])
        else:
            best = None
            for _ in range(6):
                tok = torch.cuda.Stream(device=device, priority=-1)
                got = admit(tok, [])
                ok = len(got) >= want and not stream_starved_by(null, tok, sleeper=got[0])
                r = ratio(tok, got) if len(got) >= 2 else float("inf")
                if best is None or r < best[2]:
                    best = (tok, got, r)
                if ok and r < 0.8:
                    break
                ent["passed_over"] += 1
            else:
                ent["compromised"] = True
            ent["tok"], ent["dense"], ent["mini_ratio"] = best
        while len(ent["dense"]) < want:      # nothing admissible left (four hardware queues: the default stream's and three more): take what there is
            if len(ent["dense"]) < 2:
                ent["compromised"] = True
            ent["unadmitted"] += 1
            ent["dense"].append(torch.cuda.Stream(device=device))
    ent["probe_s"] += time.perf_counter() - t0
    if ent["compromised"] and not ent.get("warned"):
        import warnings
        ent["warned"] = True
        warnings.warn("point_sam_amd: no set of HIP streams on this process's hardware queues lets the pipeline's stages overlap (GPU_MAX_HW_QUEUES too small, or "
                      "every queue shares a command-processor pipe with another): pipelines will run, at up to 2x the time per batch -- pipeline_streams_report()",
                      RuntimeWarning, stacklevel=2)
    return ent["tok"], list(ent["dense"][:dense])


def side_stream(device):
    """A normal-priority stream for work beside the pipelines (the per-step RCCL gather of the results, dist.SideStreamGather): not on a dense stream's
    hardware queue and NOT starved by the tokenizer stream.  With four hardware queues per priority the one queue left beside the default stream's and the two
    dense streams' is, as a rule, exactly the tokenizer queue's pipe partner -- a collective on it would sit until the tokenizer stream's pending wait resolves
    (an earlier batch's dense stage: ~10 ms) -- so a candidate that shares the DEFAULT stream's queue is accepted: the gather is ordered after the caller's wait
    for the batch it gathers anyway.  Falls back to a fresh stream when nothing better is found or no pipeline streams exist yet."""
    if os.environ.get("PSAM_PRIVATE_STREAMS", "0") == "1" or not hasattr(torch.cuda, "_sleep"):
        return torch.cuda.Stream(device=device)
    with _LOCK:
        dev = torch.device(device)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        ent = _POOL.get((dev.type, idx))
        if ent is None or ent["tok"] is None:
            return torch.cuda.Stream(device=device)
        if ent.get("side") is not None:
            return ent["side"]
        t0 = time.perf_counter()
        with torch.cuda.device(idx):
            null = torch.cuda.default_stream(idx)
            last = None
            for _ in range(10):
                c = last = torch.cuda.Stream(device=device)
                if any(streams_alias(c, d) for d in ent["dense"][:2]):      # (the probed pair; a third, unprobed dense stream may sit anywhere)
                    continue
                if streams_alias(c, null) or not stream_starved_by(c, ent["tok"]):
                    ent["side"] = c
                    break
            else:
                ent["side"] = last
                ent["side_unprobed"] = True
        ent["probe_s"] += time.perf_counter() - t0
        return ent["side"]


def pipeline_streams_report():
    """What the pool holds and what admission cost, per device (bench line: `config.streams`)."""
    return {f"{k[0]}:{k[1]}": {"tokenizer_stream_id": v["tok"].stream_id if v["tok"] is not None else None, "dense_stream_ids": [d.stream_id for d in v["dense"]],
                               "tokenizer_candidates_passed_over": v["passed_over"], "dense_candidates_seen": len(v["cands"]), "compromised": v["compromised"], "dense_streams_beyond_the_probed_set": v["unadmitted"],
                               "mini_pipeline_ratio": None if v["mini_ratio"] is None else round(v["mini_ratio"], 3), "probe_seconds": round(v["probe_s"], 3),
                               "side_stream_id": v["side"].stream_id if v.get("side") is not None else None}
            for k, v in _POOL.items()}
